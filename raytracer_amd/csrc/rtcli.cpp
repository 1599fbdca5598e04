// rtcli — C++ mirror of `cabal run raytracer <config.json>` (/root/reference/app/Main.hs:13-59 +
// createPPM, app/Rendering/ImageGenerator.hs:33-128) for hosts without GHC: same banners, same
// timestamped out/<%Y%m%d-%H%M%S>.ppm, same "Time taken to create and write file" line, but the
// worker fan-out is one rt_render call into the MI355X render core.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <sys/stat.h>
#include <vector>

#include "rthost.h"

int main(int argc, char **argv) {
  printf("Initializing...\n");
  if (argc != 2) { printf("Usage: rtcli <config-file>\n"); return 0; }      // Main.hs:27-28 (exit code 0, as the reference)
  rth_config *cfg = nullptr;
  if (rth_config_load(argv[1], &cfg) != RT_OK) { fprintf(stderr, "%s\n", rth_last_error()); printf("Failed to load configuration.\n"); return 1; }   // :24-26
  rth_config_view v;
  rth_config_get(cfg, &v);
  printf("Configuration loaded: %dx%d, %d spp, maxBounces %d, bvhMaxDepth %d, leafThreshold %d\n", v.params.width, v.params.height,
         v.params.samples_per_pixel, v.params.max_bounces, v.bvh_max_depth, v.leaf_threshold);
  mkdir("out", 0755);
  char name[128];
  time_t now = time(nullptr);
  struct tm tmv;
  localtime_r(&now, &tmv);
  strftime(name, sizeof name, "out/%Y%m%d-%H%M%S.ppm", &tmv);                 // Main.hs:48-50
  auto t0 = std::chrono::steady_clock::now();

  printf("[Scene Setup] Starting scene parsing and BVH construction...\n");   // ImageGenerator.hs:39
  printf("[Scene Info] Scene has %d triangles.\n", v.n_triangles);            // SceneParser.hs:75
  rth_bvh *bvh = nullptr;
  if (rth_bvh_build(v.n_objs, v.obj_kind, v.obj_data, v.obj_material, v.leaf_threshold, v.bvh_max_depth, 8, &bvh) != RT_OK) {
    fprintf(stderr, "%s\n", rth_last_error()); return 1;
  }
  printf("[Scene Setup] BVH construction complete.\n");
  printf("[Scene Setup] Scene and BVH complete.\n");
  rt_scene_desc desc;
  memset(&desc, 0, sizeof desc);
  rth_bvh_fill_desc(bvh, &desc);
  desc.n_materials = v.n_materials; desc.materials = v.materials;
  desc.n_lights = v.n_lights; desc.lights = v.lights;
  desc.bg_kind = v.bg_kind;
  memcpy(desc.bg_color1, v.bg_color1, sizeof desc.bg_color1); memcpy(desc.bg_color2, v.bg_color2, sizeof desc.bg_color2);
  printf("[Scene Setup] Starting sky sphere loading...\n");
  void *sky = nullptr;
  if (v.sky_path) {
    if (rth_sky_load(v.sky_path, &desc.sky_format, &desc.sky_width, &desc.sky_height, &sky) != RT_OK) { fprintf(stderr, "%s\n", rth_last_error()); return 1; }
    desc.sky_pixels = sky;
  }
  printf("[Scene Setup] Sky sphere loaded.\n");
  printf("[Scene Setup] Camera created, lights are set up.\n");
  // RT_DEVICES: the device group of the single blocking rt_render below, as the Haskell binding selects it
  // (integration/Rendering/GpuRender.hs: rt_init_devices 0 NULL = every visible GPU).  "all" or a comma-separated id list
  // (ids may repeat: virtual devices on one GPU); unset = the current device only.
  if (const char *devs = getenv("RT_DEVICES")) {
    std::vector<int32_t> ids;
    if (strcmp(devs, "all") != 0)
      for (const char *q = devs; *q;) { ids.push_back((int32_t)strtol(q, (char **)&q, 10)); if (*q == ',') ++q; else if (*q) break; }
    if (rt_init_devices((int32_t)ids.size(), ids.empty() ? nullptr : ids.data()) != RT_OK) { fprintf(stderr, "rtcore: %s\n", rt_last_error()); return 1; }
    printf("[Rendering] device group of %d\n", rt_device_count());
  }
  rt_scene *scene = nullptr;
  if (rt_scene_create(&desc, &scene) != RT_OK) { fprintf(stderr, "rtcore: %s\n", rt_last_error()); return 1; }
  printf("[Rendering] Launching the MI355X render core...\n");
  const size_t npx = (size_t)v.params.width * v.params.height;
  std::vector<uint8_t> rgb(npx * 3);
  rt_output out;
  memset(&out, 0, sizeof out);
  out.rgb8 = rgb.data();
  rt_stats st;
  const char *seed = getenv("RT_SEED");
  if (seed) v.params.seed = strtoull(seed, nullptr, 10);
  if (rt_render(scene, &v.camera, &v.params, &out, &st) != RT_OK) { fprintf(stderr, "rtcore: %s\n", rt_last_error()); return 1; }
  printf("[Rendering] kernel %.3f ms, %.1f Mpaths/s\n", st.ms_render, st.paths / (st.ms_render * 1e3));
  if (rth_write_ppm(name, v.params.width, v.params.height, rgb.data()) != RT_OK) { fprintf(stderr, "%s\n", rth_last_error()); return 1; }
  printf("[Done] Raytracing ended. Finishing up...\n");                        // ImageGenerator.hs:128
  double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  printf("Image saved to %s\n", name);
  printf("Time taken to create and write file: %.2f seconds\n", secs);        // Main.hs:58
  rt_scene_destroy(scene);
  rth_bvh_destroy(bvh);
  rth_sky_free(sky);
  rth_config_destroy(cfg);
  return 0;
}
