// examples/global_illumination.cpp -- the reference program examples/global_illumination/main.adb
// (lines 29-74 and 149-161) restated with the C++ mirror of Madarch's packages: same scene data,
// same call order, minus the window loop.  Usage: global_illumination W H FRAMES [out.f32 [out.ppm]]
#include "madarch.hpp"

#include <cstdio>
#include <cstdlib>

using namespace Madarch;

int main(int argc, char **argv)
{
   const int W = argc > 1 ? atoi(argv[1]) : 1000, H = argc > 2 ? atoi(argv[2]) : 1000, frames = argc > 3 ? atoi(argv[3]) : 1;
   try {
      Scenes::Scene Scene = Scenes::Compile({{Primitives::Spheres::Sphere, 20}, {Primitives::Planes::Plane, 10}, {Primitives::Boxes::Box, 10}},
                                            {{Lights::Spot_Lights::Spot_Light, 4}}, Scenes::Partitioning_Settings{false});
      Windows::Window Window = Windows::Open(W, H, "Global_Illumination");
      Renderers::Renderer Renderer = Renderers::Create(Window, Scene, {}, Renderers::No_Volumetrics);

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
      std::vector<float> image = Renderer.Read_Framebuffer();
      if (argc > 4) {
         FILE *out = fopen(argv[4], "wb");
         if (!out) return 2;
         fwrite(image.data(), sizeof(float), image.size(), out);
         fclose(out);
      }
      if (argc > 5) { // the window's pixels of the last frame as a binary PPM
         const uint8_t *px = Renderer.Front_Buffer();
         FILE *out = fopen(argv[5], "wb");
         if (!out) return 2;
         fprintf(out, "P6\n%d %d\n255\n", W, H);
         for (size_t i = 0; i < (size_t)W * H; ++i) fwrite(px + 4 * i, 1, 3, out);
         fclose(out);
      }
      double sum = 0;
      for (float v : image) sum += (v == v) ? v : 0;
      printf("global_illumination %dx%d frames %d mean %.6f\n", W, H, frames, sum / image.size());
   } catch (const std::exception &e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }
   return 0;
}
