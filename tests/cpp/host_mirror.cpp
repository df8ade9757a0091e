// host_mirror.cpp -- drives libmi355rt.so through the C++ host mirror (include/mi355rt.hpp) the way
// RayTracer.java / RayTracerInterface.java drive the reference; the pytest wrappers compare its output with the
// oracle (tests/test_cpp_host.py).
//   host_mirror abi
//   host_mirror render <mesh.ply> <width> <height> <multiSamples> <seed> <fov> <out.bin> [ranks]
//       ranks > 0: the scene gets a device list that names GPU 0 that many times (Scene::setDevices: the N-rank path on one GPU)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "mi355rt.hpp"

using namespace mi355rt;

static int run_abi()
{
   std::printf("abi %d devices %d\n", mi355rt_abi_version(), mi355rt_device_count());
   Light light({ 0.f, 100.f, 100.f }, { 0.3f, 0.3f, 0.3f }, { 0.5f, 0.5f, 0.9f });
   ColorInformation white({ 1.f, 1.f, 1.f });
   ReflectionMaterial mirror({ 0.9f, 0.9f, 1.0f });
   BRDFMaterial felt(3, { 0.2f, 0.8f, 0.3f });
   if (white.abi().shininess != 20.0f || mirror.abi().shininess != 100.0f || felt.abi().brdf_id != 3 ||
       mirror.abi().kind != MI355RT_MAT_REFLECTION || felt.abi().kind != MI355RT_MAT_BRDF) {
      std::printf("material defaults wrong\n");
      return 1;
   }
   RefractionMaterial glass({ 0.5f, 0.5f, 0.5f }, RefractionMaterial::INDEX_OF_GLASS);
   CheckerboardMaterial checker({ 1.f, 1.f, 0.f }, { 1.f, 0.f, 0.f }, 10.f, 10.f, 10.f);
   if (glass.abi().kind != MI355RT_MAT_REFRACTION || glass.abi().shininess != 100.0f || glass.abiExt().refraction_index != 1.52f ||
       checker.abi().kind != MI355RT_MAT_CHECKERBOARD || checker.abi().diffuse[0] != 0.0f || checker.abi().shininess != 20.0f ||
       checker.abiExt().checker_cell[2] != 10.f || checker.abiExt().checker_color1[1] != 1.f) {
      std::printf("refraction / checkerboard defaults wrong\n");
      return 1;
   }
   try {
      Plane two({ { 0.f, 0.f, 0.f }, { 1.f, 0.f, 0.f } }, white);
      std::printf("no error for a two-corner plane\n");
      return 1;
   } catch (const std::invalid_argument &) {
   }
   try {
      TriangleMesh missing("/nonexistent/mesh.ply", white, true);
      std::printf("no error for a missing file\n");
      return 1;
   } catch (const Error &e) {
      if (e.code != MI355RT_E_IO) {
         std::printf("wrong code %d\n", e.code);
         return 1;
      }
      std::printf("missing file -> E_IO: %s\n", e.what());
   }
   // without a HIP device the library must refuse, not fall back
   if (mi355rt_device_count() == 0) {
      TriangleMesh tri({ 0, 0, 0, 1, 0, 0, 0, 1, 0 }, { 0, 1, 2 }, white, true);
      PhongLightingModel model(light, { &tri });
      try {
         Scene scene({ &tri }, light, { 0, 0, 0 }, model, 15.0f);
         std::printf("scene built without a device\n");
         return 1;
      } catch (const Error &e) {
         if (e.code != MI355RT_E_DEVICE) return 1;
         std::printf("no device -> E_DEVICE\n");
      }
   }
   std::printf("ABI_OK\n");
   return 0;
}

static int write_frame(const Camera &camera, int W, int H, const char *path)
{
   FILE *f = std::fopen(path, "wb");
   if (!f) return 1;
   const float header[8] = { camera.cam.origin[0], camera.cam.origin[1], camera.cam.origin[2], camera.cam.viewport_w, camera.cam.viewport_h,
                             RTStatics::EPSILON(), (float) W, (float) H };
   std::fwrite(header, sizeof(float), 8, f);
   std::fwrite(camera.rowMajorPixels().data(), sizeof(float), (size_t) 3 * W * H, f);
   std::fwrite(camera.hitId.data(), sizeof(int32_t), (size_t) W * H, f);
   std::fwrite(camera.hitW.data(), sizeof(float), (size_t) W * H, f);
   std::fwrite(camera.hitObj.data(), sizeof(int32_t), (size_t) W * H, f);
   std::fwrite(camera.image.data(), sizeof(uint32_t), (size_t) W * H, f);   // Camera.image
   std::fclose(f);
   std::printf("RENDER_OK %dx%d pixel(0,0)=%g\n", W, H, camera.getPixel(0, 0)[0]);
   return 0;
}

// Scenes.WHITTED_SCENE as RayTracer.getScene / getWhittedObjects(false) build it (RayTracer.java:111-113,155-167)
static int run_whitted(char **argv)
{
   const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), ms = std::atoi(argv[4]);
   const unsigned long long seed = std::strtoull(argv[5], nullptr, 10);
   Light light({ 0.f, 100.f, 100.f }, { 0.3f, 0.3f, 0.3f }, { 0.5f, 0.5f, 0.9f });
   Plane plane({ { -50.f, 0.f, -100.f }, { -50.f, -40.f, 25.f }, { 50.f, -40.f, 25.f }, { 50.f, 0.f, -100.f } },
               CheckerboardMaterial({ 1.f, 1.f, 0.f }, { 1.f, 0.f, 0.f }, 10.f, 10.f, 10.f));
   ReflectionMaterial blue({ 0.f, 0.f, 1.f });
   RefractionMaterial glass({ 128 / 255.f, 128 / 255.f, 128 / 255.f }, RefractionMaterial::INDEX_OF_GLASS);
   Sphere sphere1(5.f, { 0.f, -12.f, 0.f }, glass);
   Sphere sphere2(3.f, { 5.f, -15.f, -10.f }, blue);
   std::vector<const BoundingVolume *> volumes { plane.getBoundingVolume(), sphere1.getBoundingVolume(), sphere2.getBoundingVolume() };
   PhongLightingModel model(light, volumes);
   Scene scene(volumes, light, { 0.f, 0.f, 0.f }, model, 35.0f);
   Camera camera(scene, ms, 1, RTStatics::NEAR_PLANE, W, H);
   camera.rngSeed = seed;
   camera.createImage();
   return write_frame(camera, W, H, argv[6]);
}

static int run_render(int argc, char **argv)
{
   const std::string ply = argv[2];
   const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), ms = std::atoi(argv[5]);
   const unsigned long long seed = std::strtoull(argv[6], nullptr, 10);
   const float fov = (float) std::atof(argv[7]);
   const int ranks = argc > 9 ? std::atoi(argv[9]) : 0;
   Light light({ 0.f, 100.f, 100.f }, { 0.3f, 0.3f, 0.3f }, { 0.5f, 0.5f, 0.9f });     // RayTracer.java:106
   TriangleMesh mesh(ply, ColorInformation({ 0.9f, 0.9f, 0.9f }), true);
   PhongLightingModel model(light, { &mesh });
   Scene scene({ &mesh }, light, { 0.f, 0.f, 0.f }, model, fov);
   if (ranks > 0) {
      scene.setDevices(std::vector<int32_t>((size_t) ranks, 0));
      const mi355rt_multi_info info = scene.multiInfo();
      std::printf("ranks %u transport %d\n", info.n_ranks, info.transport);
      if ((int) info.n_ranks != ranks) return 1;
   }
   const std::array<double, 7> bs = scene.buildSeconds();
   std::printf("build seconds %.4f %.4f %.4f %.4f %.4f %.4f %.4f\n", bs[0], bs[1], bs[2], bs[3], bs[4], bs[5], bs[6]);
   Camera camera(scene, ms, 1, RTStatics::NEAR_PLANE, W, H);
   camera.rngSeed = seed;
   int fired = 0;
   camera.addActionListener([&](int id, const char *what) { fired += (id == 1 && std::strcmp(what, "finished") == 0) ? 1 : 0; });
   camera.createImage();
   if (fired != 1) {
      std::printf("listener fired %d times\n", fired);
      return 1;
   }
   return write_frame(camera, W, H, argv[8]);
}

int main(int argc, char **argv)
{
   try {
      if (argc >= 2 && std::strcmp(argv[1], "abi") == 0) return run_abi();
      if (argc >= 9 && std::strcmp(argv[1], "render") == 0) return run_render(argc, argv);
      if (argc >= 7 && std::strcmp(argv[1], "whitted") == 0) return run_whitted(argv);
   } catch (const std::exception &e) {
      std::printf("exception: %s\n", e.what());
      return 2;
   }
   std::printf("usage: host_mirror abi | render <ply> <W> <H> <ms> <seed> <fov> <out.bin> | whitted <W> <H> <ms> <seed> <out.bin>\n");
   return 64;
}
