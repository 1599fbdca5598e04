/* Prints sizeof / offsetof of every struct of include/rtcore.h, one "name value" per line.
 * tests/test_haskell_offsets.py compares integration/Rendering/GpuRender.hs (and raytracer_amd/_ffi.py) with this. */
#include <stddef.h>
#include <stdio.h>
#include "rtcore.h"
#define S(t) printf("sizeof_%s %zu\n", #t, sizeof(t))
#define O(t, f) printf("off_%s_%s %zu\n", #t, #f, offsetof(t, f))
int main(void) {
  S(rt_material); O(rt_material, id); O(rt_material, flags); O(rt_material, diffuse); O(rt_material, emission);
  O(rt_material, shininess); O(rt_material, ior); O(rt_material, transmission);
  S(rt_light); O(rt_light, kind); O(rt_light, reserved); O(rt_light, vec); O(rt_light, intensity);
  S(rt_scene_desc); O(rt_scene_desc, n_nodes); O(rt_scene_desc, node_bounds); O(rt_scene_desc, node_child);
  O(rt_scene_desc, n_prims); O(rt_scene_desc, prim_kind); O(rt_scene_desc, prim_data); O(rt_scene_desc, prim_material);
  O(rt_scene_desc, prim_id); O(rt_scene_desc, n_materials); O(rt_scene_desc, materials); O(rt_scene_desc, n_lights);
  O(rt_scene_desc, lights); O(rt_scene_desc, bg_kind); O(rt_scene_desc, bg_color1); O(rt_scene_desc, bg_color2);
  O(rt_scene_desc, sky_format); O(rt_scene_desc, sky_width); O(rt_scene_desc, sky_height); O(rt_scene_desc, sky_pixels);
  S(rt_camera); O(rt_camera, origin); O(rt_camera, lower_left); O(rt_camera, horizontal); O(rt_camera, vertical);
  O(rt_camera, u); O(rt_camera, v); O(rt_camera, w); O(rt_camera, aperture);
  S(rt_params); O(rt_params, width); O(rt_params, height); O(rt_params, samples_per_pixel); O(rt_params, antialiasing);
  O(rt_params, max_bounces); O(rt_params, rr_start_bounce); O(rt_params, exposure); O(rt_params, gamma); O(rt_params, seed);
  O(rt_params, collect_stats); O(rt_params, box_precision);
  S(rt_output); O(rt_output, radiance); O(rt_output, color); O(rt_output, rgb8); O(rt_output, prim_id);
  S(rt_stats); O(rt_stats, ms_render); O(rt_stats, ms_total); O(rt_stats, paths); O(rt_stats, rays_closest); O(rt_stats, rays_shadow);
  printf("RT_ABI_VERSION %d\n", RT_ABI_VERSION);
  return 0;
}
