// examples/global_illumination.cpp -- the reference program examples/global_illumination/main.adb
// (lines 29-74 and 149-161) restated with the C++ mirror of Madarch's packages: same scene data,
// same call order, minus the window loop.  Usage: global_illumination W H FRAMES [out.f32 [out.ppm]]
//                                                  [--rank R --world N --id-file PATH [--device D]]
//                                                  [--rank R --world N --peer-dir DIR [--device D]]
// With --world N the program is one of N processes, one per GPU of a node (device = rank unless --device says
// otherwise): the renderers join a communicator inside libmadarch_hip.so, every Render is one frame of the sharded
// schedule (probe slices + RCCL all-gather + interleaved screen tiles), and rank 0 gathers the tiles of the last
// frame before it writes the image.  No other runtime is involved: the 128-byte id travels through --id-file.
#include "madarch.hpp"

#include <cstdio>
#include <cstdlib>
#include <string>

using namespace Madarch;

int main(int argc, char **argv)
{
   int rank = 0, world = 1, device = -1;
   std::string id_file, peer_dir;
   int nargs = argc;
   for (int i = 1; i < argc; ++i) // the options go last: everything before them is positional
      if (argv[i][0] == '-' && argv[i][1] == '-') { nargs = i; break; }
   for (int i = nargs; i + 1 < argc; i += 2) {
      const std::string o = argv[i];
      if (o == "--rank") rank = atoi(argv[i + 1]);
      else if (o == "--world") world = atoi(argv[i + 1]);
      else if (o == "--device") device = atoi(argv[i + 1]);
      else if (o == "--id-file") id_file = argv[i + 1];
      else if (o == "--peer-dir") peer_dir = argv[i + 1];
      else { fprintf(stderr, "unknown option %s\n", argv[i]); return 2; }
   }
   argc = nargs;
   if (world > 1 && id_file.empty() && peer_dir.empty()) { fprintf(stderr, "--world needs --id-file (RCCL) or --peer-dir (the peer exchange)\n"); return 2; }
   const int W = argc > 1 ? atoi(argv[1]) : 1000, H = argc > 2 ? atoi(argv[2]) : 1000, frames = argc > 3 ? atoi(argv[3]) : 1;
   try {
      Scenes::Scene Scene = Scenes::Compile({{Primitives::Spheres::Sphere, 20}, {Primitives::Planes::Plane, 10}, {Primitives::Boxes::Box, 10}},
                                            {{Lights::Spot_Lights::Spot_Light, 4}}, Scenes::Partitioning_Settings{false});
      Windows::Window Window = Windows::Open(W, H, "Global_Illumination");
      Renderers::Renderer Renderer = Renderers::Create(Window, Scene, {}, Renderers::No_Volumetrics, device >= 0 ? device : rank);
      if (!id_file.empty()) Renderer.Join_Node(rank, world, id_file);
      // --peer-dir: the same sharded frame with the peer exchange (device-to-device copies between the ranks' processes; the ranks
      // may share one GPU: --device 0).  Every rank then writes ITS tiles (out.f32.rank<R>): their sum is the frame.
      else if (!peer_dir.empty()) Renderer.Join_Peers(rank, world, peer_dir);

      Entities::Entity Spot_Light_Instance = Lights::Spot_Lights::Create({3.5f, 5.0f, 2.0f}, {1.0f, 0.0f, 0.0f}, 3.1415f / 4.0f, {0.9f, 0.9f, 0.8f});
      Materials::Id Wall_Mat_1 = Renderer.Add_Material(Materials::Create({0.0f, 0.0f, 0.0f}, 0.0f, 0.6f));
      Materials::Id Wall_Mat_2 = Renderer.Add_Material(Materials::Create({1.0f, 0.0f, 0.0f}, 0.0f, 0.6f));
      Materials::Id Wall_Mat_3 = Renderer.Add_Material(Materials::Create({0.0f, 0.0f, 1.0f}, 0.0f, 0.6f));
      Materials::Id Sphere_Mat = Renderer.Add_Material(Materials::Create({0.1f, 0.1f, 0.1f}, 0.9f, 0.1f));
      Materials::Id Box_Mat = Renderer.Add_Material(Materials::Create({0.0f, 1.0f, 0.0f}, 0.8f, 0.3f));

      const Entities::Entity Planes[] = {
         Primitives::Planes::Create({0.0f, 1.0f, 0.0f}, 1.0f, Wall_Mat_1), Primitives::Planes::Create({0.0f, -1.0f, 0.0f}, 7.0f, Wall_Mat_1),
         Primitives::Planes::Create({1.0f, 0.0f, 0.0f}, 1.0f, Wall_Mat_2), Primitives::Planes::Create({-1.0f, 0.0f, 0.0f}, 7.0f, Wall_Mat_3),
         Primitives::Planes::Create({0.0f, 0.0f, 1.0f}, 6.0f, Wall_Mat_1), Primitives::Planes::Create({0.0f, 0.0f, -1.0f}, 7.0f, Wall_Mat_1)};
      for (auto &Plane : Planes) Renderer.Add_Primitive(Primitives::Planes::Plane, Plane);
      Renderer.Add_Primitive(Primitives::Spheres::Sphere, Primitives::Spheres::Create({3.0f, 4.0f, 3.0f}, 1.0f, Sphere_Mat));
      Renderer.Add_Primitive(Primitives::Boxes::Box, Primitives::Boxes::Create({3.0f, 0.0f, 4.0f}, {1.5f, 1.5f, 1.5f}, Box_Mat));
      Renderer.Set_Camera_Position({2.0f, 2.0f, 0.0f});
      Renderer.Set_Light(1, Lights::Spot_Lights::Spot_Light, Spot_Light_Instance);

      for (int f = 0; f < frames; ++f) {
         Renderer.Render();
         Renderer.Swap_Buffers(); // renderers.adb:320, here into pinned host memory; no wait
      }
      if (!peer_dir.empty()) {
         Renderer.Finish();
         std::vector<float> mine = Renderer.Read_Framebuffer();
         if (argc > 4) {
            const std::string path = std::string(argv[4]) + ".rank" + std::to_string(rank);
            FILE *out = fopen(path.c_str(), "wb");
            if (!out) return 2;
            fwrite(mine.data(), sizeof(float), mine.size(), out);
            fclose(out);
         }
         Renderers::Renderer::Peers_Barrier(rank, world, peer_dir, "done"); // nobody's atlases go while a peer may still copy out of them
         Renderer.Leave_Node();
         printf("global_illumination %dx%d frames %d rank %d of %d (peer exchange)\n", W, H, frames, rank, world);
         return 0;
      }
      if (!id_file.empty()) {
         Renderer.Gather_Frame(0); // every rank's tiles into rank 0's framebuffer
         if (rank != 0) { Renderer.Barrier(); Renderer.Leave_Node(); return 0; }
      }
      std::vector<float> image = Renderer.Read_Framebuffer();
      if (argc > 4) {
         FILE *out = fopen(argv[4], "wb");
         if (!out) return 2;
         fwrite(image.data(), sizeof(float), image.size(), out);
         fclose(out);
      }
      if (argc > 5 && world == 1) { // the window's pixels of the last frame as a binary PPM (a rank's window holds its own tiles only)
         const uint8_t *px = Renderer.Front_Buffer();
         FILE *out = fopen(argv[5], "wb");
         if (!out) return 2;
         fprintf(out, "P6\n%d %d\n255\n", W, H);
         for (size_t i = 0; i < (size_t)W * H; ++i) fwrite(px + 4 * i, 1, 3, out);
         fclose(out);
      }
      double sum = 0;
      for (float v : image) sum += (v == v) ? v : 0;
      printf("global_illumination %dx%d frames %d ranks %d mean %.6f\n", W, H, frames, world, sum / image.size());
      if (!id_file.empty()) { Renderer.Barrier(); Renderer.Leave_Node(); }
   } catch (const std::exception &e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }
   return 0;
}
