// examples/light_shafts.cpp -- the reference program examples/light_shafts/main.adb:29-59,140-155
// restated with the C++ mirror (volumetrics on, default settings).  Usage: light_shafts W H FRAMES [out.f32 [out.ppm]]
#include "madarch.hpp"

#include <cstdio>
#include <cstdlib>

using namespace Madarch;

int main(int argc, char **argv)
{
   const int W = argc > 1 ? atoi(argv[1]) : 1000, H = argc > 2 ? atoi(argv[2]) : 1000, frames = argc > 3 ? atoi(argv[3]) : 1;
   try {
      Scenes::Scene Scene = Scenes::Compile({{Primitives::Spheres::Sphere, 20}, {Primitives::Planes::Plane, 10}, {Primitives::Boxes::Box, 10}},
                                            {{Lights::Point_Lights::Point_Light, 4}}, Scenes::Partitioning_Settings{false});
      Renderers::Renderer Renderer = Renderers::Create(Windows::Open(W, H, "Light_Shafts"), Scene);
      Entities::Entity Point_Light_Instance = Lights::Point_Lights::Create({5.0f, 3.0f, 6.0f}, {0.9f, 0.9f, 0.9f});
      const Entities::Entity Planes[] = {
         Primitives::Planes::Create({0.0f, 1.0f, 0.0f}, 1.0f, 0), Primitives::Planes::Create({0.0f, -1.0f, 0.0f}, 7.0f, 0),
         Primitives::Planes::Create({1.0f, 0.0f, 0.0f}, 1.0f, 1), Primitives::Planes::Create({-1.0f, 0.0f, 0.0f}, 7.0f, 2),
         Primitives::Planes::Create({0.0f, 0.0f, 1.0f}, 6.0f, 0), Primitives::Planes::Create({0.0f, 0.0f, -1.0f}, 7.0f, 0)};
      for (auto &Plane : Planes) Renderer.Add_Primitive(Primitives::Planes::Plane, Plane);
      Renderer.Add_Primitive(Primitives::Spheres::Sphere, Primitives::Spheres::Create({3.0f, 4.0f, 3.0f}, 1.0f, 3));
      Renderer.Add_Primitive(Primitives::Boxes::Box, Primitives::Boxes::Create({3.0f, 0.0f, 4.0f}, {1.5f, 1.5f, 1.5f}, 2));
      Renderer.Set_Material(0, Materials::Create({0.0f, 0.0f, 0.0f}, 0.0f, 1.0f));
      Renderer.Set_Material(1, Materials::Create({1.0f, 0.0f, 0.0f}, 0.0f, 1.0f));
      Renderer.Set_Material(2, Materials::Create({0.0f, 1.0f, 0.0f}, 0.0f, 1.0f));
      Renderer.Set_Material(3, Materials::Create({0.0f, 0.0f, 1.0f}, 0.0f, 1.0f));
      Renderer.Set_Camera_Position({2.0f, 2.0f, 0.0f});
      Renderer.Set_Light(1, Lights::Point_Lights::Point_Light, Point_Light_Instance);

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
      printf("light_shafts %dx%d frames %d mean %.6f\n", W, H, frames, sum / image.size());
   } catch (const std::exception &e) {
      fprintf(stderr, "error: %s\n", e.what());
      return 1;
   }
   return 0;
}
