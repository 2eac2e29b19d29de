// madarch.hpp -- C++ host-side mirror of Madarch's Ada packages over the C ABI of
// madarch_hip.h (header only).
//
// The reference's host language is Ada (no Ada toolchain exists in this pipeline), so the
// host side above the C ABI is restated in C++ package for package, with the Ada names:
//   Madarch::Values, Components, Entities, Materials, Primitives::{Spheres, Planes, Boxes,
//   Triangles}, Lights::{Point_Lights, Spot_Lights}, GPU_Types, Scenes, Windows, Renderers.
// Reference files: madarch/madarch-*.ads, madarch/support/gpu_types-*.adb (cited per item).
// Ada exceptions map to Program_Error / Constraint_Error below.  examples/*.cpp restate the
// three example programs with it; madarch_amd/ is the same mirror in Python.
#pragma once

#include "madarch_hip.h"

#include <cstdlib>
#include <algorithm>
#include <sys/stat.h>
#include <atomic>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace Madarch {

struct Program_Error : std::runtime_error { using std::runtime_error::runtime_error; };
struct Constraint_Error : std::runtime_error { using std::runtime_error::runtime_error; };

using Single = float;  // GL.Types.Single
using Int = int32_t;   // GL.Types.Int
using Vector3 = std::array<Single, 3>;                 // Singles.Vector3
using Matrix3 = std::array<std::array<Single, 3>, 3>;  // Singles.Matrix3, [row][column]

// ---------------------------------------------------------------- madarch-values.ads:8-27
namespace Values {
enum Value_Kind { Vector3_Kind = 0, Float_Kind = 1, Int_Kind = 2 };
struct Value {
   Value_Kind Kind = Float_Kind;
   Vector3 Vector3_Value{};
   Single Float_Value = 0;
   Int Int_Value = 0;
};
inline Value Vector3(const Madarch::Vector3 &x) { Value v; v.Kind = Vector3_Kind; v.Vector3_Value = x; return v; }
inline Value Float(Single x) { Value v; v.Kind = Float_Kind; v.Float_Value = x; return v; }
inline Value Int(Madarch::Int x) { Value v; v.Kind = Int_Kind; v.Int_Value = x; return v; }
} // namespace Values

// ------------------------------------------------------------- madarch-components.ads
namespace Components {
struct Component_Internal { std::string Name; Values::Value_Kind Kind; };
using Component = std::shared_ptr<const Component_Internal>; // compared by identity, as the access type is
inline Component Create(const std::string &Name, Values::Value_Kind Kind) { return std::make_shared<Component_Internal>(Component_Internal{Name, Kind}); }
inline const std::string &Get_Name(const Component &C) { return C->Name; }
inline Values::Value_Kind Get_Kind(const Component &C) { return C->Kind; }
using Component_Array = std::vector<Component>;
} // namespace Components

// ------------------------------------------------------------- madarch-entities.adb:2-43
namespace Entities {
class Entity {
 public:
   using Item = std::pair<Components::Component, Values::Value>;
   Entity() = default;
   explicit Entity(std::vector<Item> values) : values_(std::make_shared<std::vector<Item>>(std::move(values))) {}
   Values::Value Get(const Components::Component &Comp) const
   {
      for (auto &cv : *values_)
         if (cv.first == Comp) return cv.second;
      throw Program_Error("Entity does not have given component.");
   }
   void Set(const Components::Component &Comp, const Values::Value &V)
   {
      for (auto &cv : *values_)
         if (cv.first == Comp) { cv.second = V; return; }
      throw Program_Error("Entity does not have given component.");
   }
   void Foreach(const std::function<void(const Components::Component &, const Values::Value &)> &Proc) const
   {
      for (auto &cv : *values_) Proc(cv.first, cv.second);
   }
 private:
   std::shared_ptr<std::vector<Item>> values_; // reference semantics, as the Ada access type
};
inline Entity Create(std::vector<Entity::Item> Values) { return Entity(std::move(Values)); }
} // namespace Entities

// ------------------------- support/gpu_types-base.ads:21-37, -structs.adb:11-38, -fixed_arrays.adb:17-39
namespace GPU_Types {
inline int Pad(int x, int amount) { while (x % amount) ++x; return x; }
inline int Alignment(Values::Value_Kind k) { return k == Values::Vector3_Kind ? 16 : 4; }
inline int Size(Values::Value_Kind k) { return k == Values::Vector3_Kind ? 12 : 4; }
// std140 struct of an entity kind (Compute_Prim_Struct_Type, madarch-scenes.adb:1272-1288)
struct Struct {
   Components::Component_Array Comps;
   int Size_Bytes() const
   {
      int total = 0;
      for (auto &c : Comps) total = Pad(total, Alignment(c->Kind)) + Size(c->Kind);
      return total;
   }
   int Offset_Of(const Components::Component &Comp) const
   {
      int off = 0;
      for (auto &c : Comps) {
         off = Pad(off, Alignment(c->Kind));
         if (c == Comp) return off;
         off += Size(c->Kind);
      }
      throw Program_Error("Index out of bounds");
   }
   int Stride() const { return Pad(Size_Bytes(), 16); }
};
// Write_Entity + Write_Value (madarch-renderers.adb:323-347) into one element image
inline std::vector<uint8_t> Element_Blob(const Struct &S, const Entities::Entity &Ent)
{
   std::vector<uint8_t> blob((size_t)S.Size_Bytes(), 0);
   Ent.Foreach([&](const Components::Component &c, const Values::Value &v) {
      int off = S.Offset_Of(c);
      if (v.Kind == Values::Vector3_Kind) std::memcpy(&blob[off], v.Vector3_Value.data(), 12);
      else if (v.Kind == Values::Float_Kind) std::memcpy(&blob[off], &v.Float_Value, 4);
      else std::memcpy(&blob[off], &v.Int_Value, 4);
   });
   return blob;
}
} // namespace GPU_Types

// ------------------------------------------------------------- madarch-materials.ads:10-25
namespace Materials {
using Id = Int;
inline const Components::Component Albedo = Components::Create("albedo", Values::Vector3_Kind);
inline const Components::Component Metallic = Components::Create("metallic", Values::Float_Kind);
inline const Components::Component Roughness = Components::Create("roughness", Values::Float_Kind);
inline Entities::Entity Create(const Vector3 &Instance_Albedo, Single Instance_Metallic, Single Instance_Roughness)
{
   return Entities::Create({{Albedo, Values::Vector3(Instance_Albedo)}, {Metallic, Values::Float(Instance_Metallic)}, {Roughness, Values::Float(Instance_Roughness)}});
}
} // namespace Materials

// ------------------------------------------------------------- madarch-primitives.ads:13-60
// A primitive KIND = name + components; the distance/normal/material functions of the four
// built-in kinds are hand-written device code in libmadarch_hip (madarch_amd/csrc).
namespace Primitives {
struct Primitive_Internal { std::string Name; Components::Component_Array Comps; };
using Primitive = std::shared_ptr<const Primitive_Internal>;
inline Primitive Create(const std::string &Name, Components::Component_Array Comps) { return std::make_shared<Primitive_Internal>(Primitive_Internal{Name, std::move(Comps)}); }
inline const std::string &Get_Name(const Primitive &P) { return P->Name; }
using Primitive_Array = std::vector<Primitive>;

namespace Materials { // madarch-primitives-materials.ads:8
inline const Components::Component Material_Id = Components::Create("material_id", Values::Int_Kind);
}
namespace Spheres { // madarch-primitives-spheres.ads:10-33
inline const Components::Component Center = Components::Create("center", Values::Vector3_Kind);
inline const Components::Component Radius = Components::Create("radius", Values::Float_Kind);
inline const Primitive Sphere = Primitives::Create("Sphere", {Center, Radius, Materials::Material_Id});
inline Entities::Entity Create(const Vector3 &C, Single R, Int Material_Id)
{
   return Entities::Create({{Center, Values::Vector3(C)}, {Radius, Values::Float(R)}, {Materials::Material_Id, Values::Int(Material_Id)}});
}
} // namespace Spheres
namespace Planes { // madarch-primitives-planes.ads:10-33
inline const Components::Component Normal = Components::Create("normal", Values::Vector3_Kind);
inline const Components::Component Offset = Components::Create("offset", Values::Float_Kind);
inline const Primitive Plane = Primitives::Create("Plane", {Normal, Offset, Materials::Material_Id});
inline Entities::Entity Create(const Vector3 &N, Single O, Int Material_Id)
{
   return Entities::Create({{Normal, Values::Vector3(N)}, {Offset, Values::Float(O)}, {Materials::Material_Id, Values::Int(Material_Id)}});
}
} // namespace Planes
namespace Boxes { // madarch-primitives-boxes.ads:10-31
inline const Components::Component Center = Components::Create("center", Values::Vector3_Kind);
inline const Components::Component Side = Components::Create("side", Values::Vector3_Kind);
inline const Primitive Box = Primitives::Create("Box", {Center, Side, Materials::Material_Id});
inline Entities::Entity Create(const Vector3 &C, const Vector3 &S, Int Material_Id)
{
   return Entities::Create({{Center, Values::Vector3(C)}, {Side, Values::Vector3(S)}, {Materials::Material_Id, Values::Int(Material_Id)}});
}
} // namespace Boxes
namespace Triangles { // madarch-primitives-triangles.ads:10-35
inline const Components::Component V1 = Components::Create("v1", Values::Vector3_Kind);
inline const Components::Component V2 = Components::Create("v2", Values::Vector3_Kind);
inline const Components::Component V3 = Components::Create("v3", Values::Vector3_Kind);
inline const Primitive Triangle = Primitives::Create("Triangle", {V1, V2, V3, Materials::Material_Id});
inline Entities::Entity Create(const Vector3 &A, const Vector3 &B, const Vector3 &C, Int Material_Id)
{
   return Entities::Create({{V1, Values::Vector3(A)}, {V2, Values::Vector3(B)}, {V3, Values::Vector3(C)}, {Materials::Material_Id, Values::Int(Material_Id)}});
}
} // namespace Triangles
} // namespace Primitives

// ------------------------------------------------------------- madarch-lights.ads:7-37
namespace Lights {
struct Light_Internal { std::string Name; Components::Component_Array Comps; };
using Light = std::shared_ptr<const Light_Internal>;
inline Light Create(const std::string &Name, Components::Component_Array Comps) { return std::make_shared<Light_Internal>(Light_Internal{Name, std::move(Comps)}); }
namespace Point_Lights { // madarch-lights-point_lights.ads:14-35
inline const Components::Component Position = Components::Create("position", Values::Vector3_Kind);
inline const Components::Component Color = Components::Create("color", Values::Vector3_Kind);
inline const Light Point_Light = Lights::Create("PointLight", {Position, Color});
inline Entities::Entity Create(const Vector3 &P, const Vector3 &C) { return Entities::Create({{Position, Values::Vector3(P)}, {Color, Values::Vector3(C)}}); }
} // namespace Point_Lights
namespace Spot_Lights { // madarch-lights-spot_lights.ads:14-41
inline const Components::Component Position = Components::Create("position", Values::Vector3_Kind);
inline const Components::Component Direction = Components::Create("direction", Values::Vector3_Kind);
inline const Components::Component Aperture = Components::Create("aperture", Values::Float_Kind);
inline const Components::Component Color = Components::Create("color", Values::Vector3_Kind);
inline const Light Spot_Light = Lights::Create("SpotLight", {Position, Direction, Aperture, Color});
inline Entities::Entity Create(const Vector3 &P, const Vector3 &D, Single A, const Vector3 &C)
{
   return Entities::Create({{Position, Values::Vector3(P)}, {Direction, Values::Vector3(D)}, {Aperture, Values::Float(A)}, {Color, Values::Vector3(C)}});
}
} // namespace Spot_Lights
} // namespace Lights

// ------------------------------------------------------------- madarch-scenes.ads:13-76
namespace Scenes {
enum Partitioning_Border_Behavior { Clamp = 0, Fallback = 1 };
enum Codegen_Loop_Strategy { Split = 0, Unify = 1 };
struct Partitioning_Settings { // scenes.ads:30-41
   bool Enable = true;
   int Index_Count = 20;
   Partitioning_Border_Behavior Border_Behavior = Clamp;
   std::array<Int, 3> Grid_Dimensions{10, 10, 20};
   Vector3 Grid_Spacing{1.0f, 1.0f, 1.0f};
   Vector3 Grid_Offset{-1.5f, -1.5f, -10.0f};
};
struct Primitive_Count { Primitives::Primitive Prim; int Count; };
struct Light_Count { Lights::Light Light; int Count; };

struct Scene_Internal {
   std::vector<Primitive_Count> Prims;
   std::vector<Light_Count> Lits;
   Partitioning_Settings Partitioning_Config;
   Single Max_Dist;
   Codegen_Loop_Strategy Loop_Strategy;
   int Kind_Index(const Primitives::Primitive &P) const
   {
      for (size_t i = 0; i < Prims.size(); ++i)
         if (Prims[i].Prim == P) return (int)i;
      throw Program_Error("primitive kind is not part of the scene");
   }
   int Kind_Index(const Lights::Light &L) const
   {
      for (size_t i = 0; i < Lits.size(); ++i)
         if (Lits[i].Light == L) return (int)i;
      throw Program_Error("light kind is not part of the scene");
   }
};
using Scene = std::shared_ptr<const Scene_Internal>;
// Scenes.Compile (scenes.ads:47-53).  The reference emits GLSL here; this back end has the
// built-in kinds as device code, so a compiled scene is its description.
inline Scene Compile(std::vector<Primitive_Count> All_Primitives, std::vector<Light_Count> All_Lights,
                     Partitioning_Settings Partitioning = {}, Single Max_Dist = 20.0f, Codegen_Loop_Strategy Loop_Strategy = Unify)
{
   return std::make_shared<Scene_Internal>(Scene_Internal{std::move(All_Primitives), std::move(All_Lights), Partitioning, Max_Dist, Loop_Strategy});
}
} // namespace Scenes

// ------------------------------------------------------------- madarch-windows.ads:12-31 (headless)
namespace Windows {
struct Window { int Width = 0, Height = 0; std::string Title; };
inline Window Open(int Width, int Height, const std::string &Title = "") { return Window{Width, Height, Title}; }
} // namespace Windows

// ------------------------------------------------------------- madarch-renderers.ads:21-97
namespace Renderers {
struct Probe_Settings { // renderers.ads:23-29
   Int Radiance_Resolution = 32, Irradiance_Resolution = 8;
   std::array<Int, 2> Probe_Count{6, 6};
   std::array<Int, 3> Grid_Dimensions{4, 3, 3};
   Vector3 Grid_Spacing{2.0f, 3.0f, 3.0f};
};
struct Volumetrics_Settings { // renderers.ads:33-41
   bool Enabled = true;
   std::array<Int, 3> Visibility_Resolution{100, 100, 100};
   Single Visibility_Step_Size = 0.1f;
   std::array<Int, 2> Scattering_Resolution{250, 250};
   Single Scattering_Step_Size = 0.1f;
};
inline const Volumetrics_Settings No_Volumetrics{false, {100, 100, 100}, 0.1f, {250, 250}, 0.1f};
enum Partitioning_Update_Method { CPU_Best = 0, CPU_Fast = 1, GPU_Fast = 2 }; // renderers.ads:93

inline void Check(int32_t status)
{
   if (status == MDH_OK) return;
   if (status == MDH_E_INDEX) throw Constraint_Error(mdh_last_error());
   throw Program_Error(mdh_last_error());
}

class Renderer {
 public:
   Renderer() = default;
   Renderer(mdh_renderer *h, Windows::Window w, Scenes::Scene s) : h_(h, [](mdh_renderer *p) { mdh_destroy(p); }), window_(std::move(w)), scene_(std::move(s)) {}
   void Render() const { Check(mdh_render(h_.get())); Check(mdh_finish(h_.get())); } // renderers.adb:302-321
   void Set_Material(Materials::Id Index, const Entities::Entity &E) const // renderers.adb:349-367
   {
      Vector3 a = E.Get(Materials::Albedo).Vector3_Value;
      Check(mdh_set_material(h_.get(), Index, a.data(), E.Get(Materials::Metallic).Float_Value, E.Get(Materials::Roughness).Float_Value));
   }
   Materials::Id Add_Material(const Entities::Entity &E) const // renderers.adb:369-377
   {
      Vector3 a = E.Get(Materials::Albedo).Vector3_Value;
      int32_t id = -1;
      Check(mdh_add_material(h_.get(), a.data(), E.Get(Materials::Metallic).Float_Value, E.Get(Materials::Roughness).Float_Value, &id));
      return id;
   }
   void Set_Primitive(const Primitives::Primitive &Prim, int Index, const Entities::Entity &E) const // renderers.adb:379-398
   {
      auto blob = GPU_Types::Element_Blob(GPU_Types::Struct{Prim->Comps}, E);
      Check(mdh_set_primitive(h_.get(), scene_->Kind_Index(Prim), Index, blob.data(), (int32_t)blob.size()));
   }
   void Add_Primitive(const Primitives::Primitive &Prim, const Entities::Entity &E) const // renderers.adb:435-456
   {
      auto blob = GPU_Types::Element_Blob(GPU_Types::Struct{Prim->Comps}, E);
      Check(mdh_add_primitive(h_.get(), scene_->Kind_Index(Prim), blob.data(), (int32_t)blob.size(), nullptr));
   }
   void Set_Light(int Index, const Lights::Light &Lit, const Entities::Entity &E) const // renderers.adb:458-483
   {
      auto blob = GPU_Types::Element_Blob(GPU_Types::Struct{Lit->Comps}, E);
      Check(mdh_set_light(h_.get(), Index, scene_->Kind_Index(Lit), blob.data(), (int32_t)blob.size()));
   }
   void Set_Camera_Position(const Vector3 &P) const { Check(mdh_set_camera_position(h_.get(), P.data())); } // renderers.adb:485-490
   void Set_Camera_Orientation(const Matrix3 &M) const // renderers.adb:492-497; [row][column] -> column-major
   {
      float cm[9];
      for (int c = 0; c < 3; ++c)
         for (int r = 0; r < 3; ++r) cm[3 * c + r] = M[r][c];
      Check(mdh_set_camera_orientation(h_.get(), cm));
   }
   Single Eval_Distance_To(const Vector3 &Position, const Primitives::Primitive_Array &Prims, Vector3 &Normal) const // renderers.adb:499-526
   {
      std::vector<int32_t> kinds;
      for (auto &p : Prims) kinds.push_back(scene_->Kind_Index(p));
      float d = 0;
      Check(mdh_eval_distance_to(h_.get(), 1, Position.data(), kinds.data(), (int32_t)kinds.size(), Normal.data(), &d));
      return d;
   }
   void Update_Partitioning(Partitioning_Update_Method Method = GPU_Fast) const { Check(mdh_update_partitioning(h_.get(), (int32_t)Method)); } // renderers.adb:757-775
   // ---- headless additions
   std::vector<float> Read_Framebuffer() const // replaces Swap_Buffers: H*W*3 floats, row 0 = top
   {
      std::vector<float> out((size_t)window_.Width * window_.Height * 3);
      Check(mdh_read_framebuffer(h_.get(), out.data()));
      return out;
   }
   // Swap_Buffers itself (renderers.adb:320): the window's RGBA8 pixels, enqueued behind the frame without a host wait ...
   void Swap_Buffers() const { Check(mdh_swap_buffers(h_.get())); }
   // ... and fetched: H*W*4 bytes owned by the renderer (valid until the second next Swap_Buffers); waits for the last swap only
   const uint8_t *Front_Buffer() const
   {
      const uint8_t *p = nullptr;
      Check(mdh_front_buffer(h_.get(), &p, nullptr));
      return p;
   }
   void Set_Option(int32_t Option, int32_t Value) const { Check(mdh_set_option(h_.get(), Option, Value)); }
   // ---- one frame on the N GPUs of a node, one process per GPU (include/madarch_hip.h, mdh_comm_*): after Join_Node,
   // Render of every rank is one frame of the sharded schedule.  The 128-byte communicator id travels from rank 0 to
   // the others through `Id_File` (written under a temporary name and renamed, so a reader never sees half of it).
   // A file left by an earlier run must never be taken for this one's: rank 0 removes it (and its temporary) BEFORE it
   // makes the new id and again once every rank has joined; the others only accept a file that is no older than their
   // own call (minus `Timeout_S`: the ranks of a run start within that) and, when the launcher hands every rank the same
   // `Nonce`, one that carries it.  The collective join itself runs under the same timeout: a rank whose peers never
   // arrive reports that and leaves the process (ncclCommInitRank cannot be cancelled) instead of hanging for ever.
   void Join_Node(int Rank, int World, const std::string &Id_File, double Timeout_S = 120.0, const std::string &Nonce = "") const
   {
      uint8_t id[MDH_COMM_ID_BYTES];
      const std::string tmp = Id_File + ".tmp";
      const auto wall0 = std::chrono::system_clock::now();
      if (Rank == 0) {
         std::remove(Id_File.c_str());
         std::remove(tmp.c_str());
         Check(mdh_comm_unique_id(id));
         FILE *f = fopen(tmp.c_str(), "wb");
         if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || (!Nonce.empty() && fwrite(Nonce.data(), 1, Nonce.size(), f) != Nonce.size())) throw Program_Error("cannot write the communicator id file");
         fclose(f);
         if (rename(tmp.c_str(), Id_File.c_str()) != 0) throw Program_Error("cannot publish the communicator id file");
      } else {
         const auto t0 = std::chrono::steady_clock::now();
         for (;;) {
            bool fresh = false;
            struct stat sb;
            if (stat(Id_File.c_str(), &sb) == 0) {
               const double age_before_call = std::chrono::duration<double>(wall0 - std::chrono::system_clock::from_time_t(sb.st_mtime)).count();
               fresh = age_before_call <= Timeout_S;
            }
            if (fresh) {
               std::vector<uint8_t> buf(sizeof id + Nonce.size() + 1);
               FILE *f = fopen(Id_File.c_str(), "rb");
               const size_t n = f ? fread(buf.data(), 1, buf.size(), f) : 0;
               if (f) fclose(f);
               if (n == sizeof id + Nonce.size() && std::equal(Nonce.begin(), Nonce.end(), buf.begin() + sizeof id)) { std::copy(buf.begin(), buf.begin() + sizeof id, id); break; }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > Timeout_S) throw Program_Error("rank 0 never published the communicator id file of this run");
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
         }
      }
      // the join under a watchdog
      std::atomic<int> state{0}; // 0 joining, 1 joined, 2 failed
      std::string error;
      mdh_renderer *h = h_.get();
      std::thread joiner([&] {
         const int32_t rc = mdh_comm_init(h, id, Rank, World);
         if (rc != 0) error = mdh_last_error();
         state = rc == 0 ? 1 : 2;
      });
      const auto t1 = std::chrono::steady_clock::now();
      while (state == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() <= Timeout_S) std::this_thread::sleep_for(std::chrono::milliseconds(5));
      if (state == 0) {
         fprintf(stderr, "rank %d: the ranks of the node did not all join within %.0f s: leaving\n", Rank, Timeout_S);
         (void)mdh_comm_abort(h);
         std::_Exit(70);
      }
      joiner.join();
      if (Rank == 0) std::remove(Id_File.c_str()); // (every rank holds the id: the collective join has returned)
      if (state == 2) throw Program_Error(error);
   }
   // ---- the same sharded frame without a collective library: the peer exchange (include/madarch_hip.h, mdh_peer_*).  Every rank
   // publishes its 512 bytes of interprocess handles as a file of `Dir` and reads the others'; Render is then the sharded frame,
   // its exchange being device-to-device copies between the ranks' processes (one process per rank, one node; several ranks
   // may share ONE GPU).  `Dir` must be this run's own (files of an earlier run are refused by their age, as in Join_Node).
   static void Publish_File(const std::string &Path, const void *Data, size_t N)
   {
      const std::string tmp = Path + ".tmp";
      FILE *f = fopen(tmp.c_str(), "wb");
      if (!f || fwrite(Data, 1, N, f) != N) throw Program_Error("cannot write " + tmp);
      fclose(f);
      if (rename(tmp.c_str(), Path.c_str()) != 0) throw Program_Error("cannot publish " + Path);
   }
   static std::vector<uint8_t> Await_File(const std::string &Path, size_t N, std::chrono::system_clock::time_point Since, double Timeout_S)
   {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
         struct stat sb;
         if (stat(Path.c_str(), &sb) == 0 && std::chrono::duration<double>(Since - std::chrono::system_clock::from_time_t(sb.st_mtime)).count() <= Timeout_S) {
            std::vector<uint8_t> buf(N + 1);
            FILE *f = fopen(Path.c_str(), "rb");
            const size_t n = f ? fread(buf.data(), 1, buf.size(), f) : 0;
            if (f) fclose(f);
            if (n == N) { buf.resize(N); return buf; }
         }
         if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > Timeout_S) throw Program_Error("a rank never published " + Path);
         std::this_thread::sleep_for(std::chrono::milliseconds(10));
      }
   }
   void Join_Peers(int Rank, int World, const std::string &Dir, double Timeout_S = 120.0) const
   {
      const auto wall0 = std::chrono::system_clock::now();
      uint8_t blob[MDH_PEER_BLOB_BYTES];
      Check(mdh_peer_export(h_.get(), blob));
      Publish_File(Dir + "/peer_" + std::to_string(Rank) + ".blob", blob, sizeof blob);
      std::vector<uint8_t> all((size_t)World * MDH_PEER_BLOB_BYTES);
      for (int q = 0; q < World; ++q) {
         const auto b = Await_File(Dir + "/peer_" + std::to_string(q) + ".blob", MDH_PEER_BLOB_BYTES, wall0, Timeout_S);
         std::copy(b.begin(), b.end(), all.begin() + (size_t)q * MDH_PEER_BLOB_BYTES);
      }
      Check(mdh_peer_init(h_.get(), all.data(), Rank, World));
   }
   // every rank has reached this point of the run (a file per rank and tag): e.g. before anybody leaves the exchange --
   // a rank's atlases must outlive the peers' last copies out of them
   static void Peers_Barrier(int Rank, int World, const std::string &Dir, const std::string &Tag, double Timeout_S = 120.0)
   {
      const auto wall0 = std::chrono::system_clock::now();
      const uint8_t one = 1;
      Publish_File(Dir + "/" + Tag + "_" + std::to_string(Rank), &one, 1);
      for (int q = 0; q < World; ++q) (void)Await_File(Dir + "/" + Tag + "_" + std::to_string(q), 1, wall0, Timeout_S);
   }
   void Finish() const { Check(mdh_finish(h_.get())); }
   void Leave_Node() const { Check(mdh_comm_destroy(h_.get())); }
   void Barrier() const { Check(mdh_comm_barrier(h_.get())); }
   double Max_Over_Ranks(double V) const { Check(mdh_comm_max_f64(h_.get(), &V)); return V; }
   // the ranks' tiles of the last frame summed into Root's framebuffer: what Root's window shows
   void Gather_Frame(int Root = 0) const { Check(mdh_comm_reduce_framebuffer(h_.get(), Root)); }
   mdh_renderer *Handle() const { return h_.get(); }
   const Windows::Window &Window() const { return window_; }
 private:
   std::shared_ptr<mdh_renderer> h_;
   Windows::Window window_;
   Scenes::Scene scene_;
};

// Renderers.Create (madarch-renderers.adb:91-300)
inline Renderer Create(const Windows::Window &Window, const Scenes::Scene &Scene, const Probe_Settings &Probes = {},
                       const Volumetrics_Settings &Volumetrics = {}, int Device = 0)
{
   std::vector<std::vector<mdh_component>> comps;
   auto decls = [&](auto &items, auto name_of) {
      std::vector<mdh_kind_decl> out;
      for (auto &it : items) {
         auto &kind = name_of(it);
         comps.emplace_back();
         for (auto &c : kind->Comps) comps.back().push_back(mdh_component{c->Name.c_str(), (int32_t)c->Kind});
         out.push_back(mdh_kind_decl{kind->Name.c_str(), it.Count, (int32_t)kind->Comps.size(), nullptr});
      }
      return out;
   };
   auto pk = decls(Scene->Prims, [](const Scenes::Primitive_Count &p) -> const Primitives::Primitive & { return p.Prim; });
   size_t npk = pk.size();
   auto lk = decls(Scene->Lits, [](const Scenes::Light_Count &l) -> const Lights::Light & { return l.Light; });
   for (size_t i = 0; i < pk.size(); ++i) pk[i].components = comps[i].data();
   for (size_t i = 0; i < lk.size(); ++i) lk[i].components = comps[npk + i].data();
   mdh_scene_desc d{};
   d.n_prim_kinds = (int32_t)pk.size(); d.prim_kinds = pk.data();
   d.n_light_kinds = (int32_t)lk.size(); d.light_kinds = lk.data();
   const auto &p = Scene->Partitioning_Config;
   d.partitioning.enable = p.Enable; d.partitioning.index_count = p.Index_Count; d.partitioning.border_behavior = p.Border_Behavior;
   for (int a = 0; a < 3; ++a) { d.partitioning.grid_dimensions[a] = p.Grid_Dimensions[a]; d.partitioning.grid_spacing[a] = p.Grid_Spacing[a]; d.partitioning.grid_offset[a] = p.Grid_Offset[a]; }
   d.max_dist = Scene->Max_Dist;
   d.loop_strategy = Scene->Loop_Strategy;
   mdh_probe_settings ps{Probes.Radiance_Resolution, Probes.Irradiance_Resolution, {Probes.Probe_Count[0], Probes.Probe_Count[1]},
                         {Probes.Grid_Dimensions[0], Probes.Grid_Dimensions[1], Probes.Grid_Dimensions[2]},
                         {Probes.Grid_Spacing[0], Probes.Grid_Spacing[1], Probes.Grid_Spacing[2]}};
   mdh_volumetrics vs{Volumetrics.Enabled ? 1 : 0, {Volumetrics.Visibility_Resolution[0], Volumetrics.Visibility_Resolution[1], Volumetrics.Visibility_Resolution[2]},
                      Volumetrics.Visibility_Step_Size, {Volumetrics.Scattering_Resolution[0], Volumetrics.Scattering_Resolution[1]}, Volumetrics.Scattering_Step_Size};
   mdh_renderer *h = nullptr;
   Check(mdh_create(Window.Width, Window.Height, &d, &ps, &vs, Device, &h));
   return Renderer(h, Window, Scene);
}
} // namespace Renderers
} // namespace Madarch
