/*
 * rthost.h — C ABI of the host-side mirror of omeldar/raytracer's scene set-up.
 *
 * The Haskell host keeps this work (north_star): BVH construction, camera set-up, config/OBJ
 * parsing and PPM output stay in Haskell.  This library exists for callers WITHOUT the Haskell
 * host — the C++ `rtcli`, bench.py and the tests — and mirrors the same functions so that a
 * scene goes through identical steps:
 *
 *   rth_bvh_build        constructBVHWithLimit        app/Hittable/BVH.hs:29-48
 *                        (chooseSplitAxis :51-71, sortByCentroid :74-80, surroundingBoxList :101-103)
 *                        + the tree -> rt_scene_desc flattening a Haskell FFI module would do
 *   rth_default_camera   defaultCamera                app/Rendering/Camera.hs:26-44
 *   rth_config_*         loadConfig / parseSceneObjects  app/Config.hs:127-141, app/Rendering/SceneParser.hs:26-80
 *   rth_write_ppm        the writer thread            app/Rendering/ImageGenerator.hs:97-115
 *
 * Same error convention as rtcore.h (0 / negative RT_ERR_*; text via rth_last_error()).
 */
#ifndef RTHOST_H
#define RTHOST_H

#include <stdint.h>
#include "rtcore.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *rth_last_error(void);

typedef struct rth_bvh rth_bvh;

/* Objects in the reference's `allObjects` order (SceneParser.hs:69-73): kind[n], data[9n]
 * (RT_PRIM_* layouts of rtcore.h), material[n].  leaf_threshold / max_depth are
 * RaytracerSettings.leafThreshold / bvhMaxDepth (Config.hs:58-64).  n_threads <= 1: serial. */
int  rth_bvh_build(int32_t n_objs, const int32_t *kind, const double *data, const int32_t *material,
                   int32_t leaf_threshold, int32_t max_depth, int32_t n_threads, rth_bvh **out);
void rth_bvh_destroy(rth_bvh *bvh);
/* n_nodes, n_leaves, depth (edges root->deepest leaf), n_prims */
void rth_bvh_info(const rth_bvh *bvh, int32_t *n_nodes, int32_t *n_leaves, int32_t *depth, int32_t *n_prims);
/* Points the node_* and prim_* fields of desc at the arrays owned by bvh (valid until destroy);
 * the other fields of desc are left untouched. */
int  rth_bvh_fill_desc(const rth_bvh *bvh, rt_scene_desc *desc);

int  rth_default_camera(const double look_from[3], const double look_at[3], const double v_up[3],
                        double vfov_deg, double aspect, double aperture, double focus_dist, rt_camera *out);

/* ---- config-file front end (mirror of loadConfig + parseSceneObjects + createPPM's set-up) ---- */
typedef struct rth_config rth_config;
/* Parses a reference-schema JSON config, loads its OBJ/MTL files, assembles
 * `allObjects`, materials and lights.  Errors mirror the reference's (`Nothing` / `error`). */
int  rth_config_load(const char *json_path, rth_config **out);
void rth_config_destroy(rth_config *cfg);
/* Filled views; pointers are owned by cfg. */
typedef struct rth_config_view {
  rt_params params;            /* image + raytracer settings; seed = 0 */
  int32_t   use_bvh, bvh_max_depth, leaf_threshold;
  rt_camera camera;            /* defaultCamera with aspect = width/height (ImageGenerator.hs:50-58) */
  int32_t   n_objs;            /* allObjects = triangles ++ configObjects (SceneParser.hs:67-73) */
  const int32_t *obj_kind; const double *obj_data; const int32_t *obj_material;
  int32_t   n_triangles;       /* "[Scene Info] Scene has N triangles." (SceneParser.hs:75) */
  int32_t   n_materials; const rt_material *materials;
  int32_t   n_lights;    const rt_light *lights;
  int32_t   bg_kind; double bg_color1[3], bg_color2[3];
  const char *sky_path;        /* NULL if none */
} rth_config_view;
int  rth_config_get(const rth_config *cfg, rth_config_view *view);

/* loadSkySphere (app/Rendering/SkySphere.hs:18-29): ".hdr" (Radiance RGBE, flat or RLE) -> RT_SKY_RGBF; any other file goes
 * through readImage + convertRGB8 there: PNG (all colour types and depths, Adam7), uncompressed 24 / 32-bit BMP, true-colour /
 * grey TGA, and binary / ASCII PPM (P6 / P3, maxval 255; not a JuicyPixels format: for tests) -> RT_SKY_RGB8, alpha dropped, grey
 * replicated, 16-bit samples cut to their high byte.  JPEG, GIF and TIFF, which JuicyPixels also decodes, return
 * RT_ERR_UNSUPPORTED.  Free pixels with rth_sky_free. */
int  rth_sky_load(const char *path, int32_t *format, int32_t *width, int32_t *height, void **pixels);
void rth_sky_free(void *pixels);

/* P3 PPM exactly as the writer thread emits it: "P3\nW H\n255\n" then one "r g b\n" per pixel. */
int  rth_write_ppm(const char *path, int32_t width, int32_t height, const uint8_t *rgb8);
/* The same bytes into memory (for tests): returns needed size; writes at most cap bytes. */
int64_t rth_format_ppm(int32_t width, int32_t height, const uint8_t *rgb8, char *out, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* RTHOST_H */
