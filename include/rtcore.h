/*
 * rtcore.h — C ABI of the MI355X render core for omeldar/raytracer.
 *
 * The reference has NO FFI/plugin interface.  The seam this ABI
 * replaces is the worker fan-out inside
 *     createPPM :: Config -> FilePath -> IO ()        app/Rendering/ImageGenerator.hs:33
 * i.e. `replicateM_ numWorkers (forkIO (worker ...))`  app/Rendering/ImageGenerator.hs:79-94
 * with its arguments (bvh, materialMap, skySphere, backgroundFunc, cameraObj,
 * sceneLights, imgW, imgH) and what `renderRow` hands back
 *                                                     app/Rendering/RenderWorker.hs:64-74.
 * A Haskell host walks its `BVHNode` tree (app/Hittable/BVH.hs:20-22) into the
 * flat arrays of rt_scene_desc, calls rt_scene_create once, rt_render per frame,
 * formats rows with its own `showPixel` and feeds its existing writer thread.
 * The binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, caller-owned buffers, no callbacks;
 *   - every entry point returns RT_OK (0) or a negative RT_ERR_* code; the text
 *     of the last error on the calling thread is rt_last_error();
 *   - all geometry/shading scalars are IEEE binary64, as in the reference
 *     (app/Core/Vec3.hs:33);
 *   - image buffers are in FILE order: index = row*width + col, row 0 = top
 *     (reference row j, `rowIdx = imgH-1-j`, app/Rendering/RenderWorker.hs:66).
 */
#ifndef RTCORE_H
#define RTCORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_ABI_VERSION 4

/* ---- status codes (error convention replaces `error`/exitFailure, app/Main.hs:24-26) */
#define RT_OK                 0
#define RT_ERR_INVALID       -1   /* bad argument / inconsistent descriptor            */
#define RT_ERR_NO_DEVICE     -2   /* no usable HIP device (the core has NO CPU fallback) */
#define RT_ERR_HIP           -3   /* a HIP runtime call failed                          */
#define RT_ERR_OOM           -4   /* host or device allocation failed                   */
#define RT_ERR_EMPTY_SCENE   -5   /* reference: "Cannot compute bounding box of empty list", app/Hittable/BVH.hs:102 */
#define RT_ERR_UNSUPPORTED   -6   /* feature outside what the kernels implement         */

/* ---- primitive kinds: the three `SceneObject`s, app/Config.hs:73-77 */
#define RT_PRIM_TRIANGLE 0   /* data = v0[3], v1[3], v2[3]            app/Hittable/Objects/Triangle.hs:13-19 */
#define RT_PRIM_SPHERE   1   /* data = center[3], radius, 5 unused    app/Hittable/Objects/Sphere.hs:11-16   */
#define RT_PRIM_PLANE    2   /* data = point[3], normal[3], 3 unused  app/Hittable/Objects/Plane.hs:11-16    */

/* ---- lights, app/Rendering/Light.hs:12-14 */
#define RT_LIGHT_POINT        0   /* vec = position  */
#define RT_LIGHT_DIRECTIONAL  1   /* vec = direction */

/* ---- background, app/Config.hs:40-43 + app/Rendering/ImageGenerator.hs:130-135 */
#define RT_BG_GRADIENT 0   /* color1 at zenith, color2 at nadir: lerp t color2 color1 */
#define RT_BG_SOLID    1   /* color1 only */

/* ---- sky sphere pixel formats, app/Rendering/SkySphere.hs:13-15 */
#define RT_SKY_NONE  0
#define RT_SKY_RGB8  1   /* uint8  rgb, row 0 = top, value/255       (LDRSkySphere) */
#define RT_SKY_RGBF  2   /* float  rgb, row 0 = top, used as is      (HDRSkySphere) */

/* ---- Material: the `Maybe` fields of app/Rendering/Material.hs:13-21 become flag bits.
 * specularColor and dissolve are never read by the integrator and do
 * not cross the ABI. */
#define RT_MAT_HAS_EMISSION     1u
#define RT_MAT_HAS_SHININESS    2u
#define RT_MAT_HAS_IOR          4u
#define RT_MAT_HAS_TRANSMISSION 8u

typedef struct rt_material {
  int32_t  id;            /* key of the reference's `Map Int Material` (Trace.hs:27-28) */
  uint32_t flags;         /* RT_MAT_HAS_* */
  double   diffuse[3];    /* diffuseColor  */
  double   emission[3];   /* emissionColor (valid iff HAS_EMISSION)  */
  double   shininess;     /* metal iff HAS_SHININESS && > 100   (Trace.hs:60-61) */
  double   ior;           /* glass iff HAS_TRANSMISSION && transmission == 1.0 && HAS_IOR (Trace.hs:48-49) */
  double   transmission;
} rt_material;

typedef struct rt_light {
  int32_t kind;           /* RT_LIGHT_* */
  int32_t reserved;
  double  vec[3];         /* position or direction */
  double  intensity[3];
} rt_light;

/* Flattened scene.  Node i carries ITS OWN box (the `AABB` field of BVHLeaf/BVHInternal,
 * app/Hittable/BVH.hs:20-22).  Node 0 is the root.
 *   internal: node_child[2i] = left index, node_child[2i+1] = right index (both > 0)
 *   leaf    : node_child[2i] = -1 - first_prim,  node_child[2i+1] = prim count (>= 1)
 * Primitives are stored in leaf order: a leaf's `DV.Vector SomeHittable` is the slice
 * [first_prim, first_prim+count) in vector order (order decides ties, HitHelpers.hs:11-16).
 * prim_id[k] = index of that object in the reference's `allObjects` list
 * (app/Rendering/SceneParser.hs:69-73); only reported back, never used for shading. */
typedef struct rt_scene_desc {
  int32_t        n_nodes;
  const double  *node_bounds;    /* 6*n_nodes: lo.xyz, hi.xyz                      */
  const int32_t *node_child;     /* 2*n_nodes                                      */
  int32_t        n_prims;
  const int32_t *prim_kind;      /* n_prims: RT_PRIM_*                             */
  const double  *prim_data;      /* 9*n_prims                                      */
  const int32_t *prim_material;  /* n_prims: material id (missing id => defaultMaterial, Trace.hs:28) */
  const int32_t *prim_id;        /* n_prims, may be NULL (then prim_id[k] = k)     */
  int32_t            n_materials;
  const rt_material *materials;
  int32_t            n_lights;
  const rt_light    *lights;
  int32_t  bg_kind;              /* RT_BG_*; ignored when a sky is given (Trace.hs:78-83) */
  double   bg_color1[3];
  double   bg_color2[3];
  int32_t  sky_format;           /* RT_SKY_* */
  int32_t  sky_width, sky_height;
  const void *sky_pixels;        /* sky_width*sky_height*3 elements */
} rt_scene_desc;

/* The `Camera` record, app/Rendering/Camera.hs:13-23 (built by defaultCamera, :26-44). */
typedef struct rt_camera {
  double origin[3];
  double lower_left[3];
  double horizontal[3];
  double vertical[3];
  double u[3], v[3], w[3];
  double aperture;
} rt_camera;

/* ImageSettings + RaytracerSettings that the hot path reads,
 * app/Config.hs:29-37,58-64.  seed keys the documented counter-based RNG
 * (Philox4x32-10) that stands in for the reference's entropy-seeded
 * generators (app/Utils/RandomHelper.hs:22). */
typedef struct rt_params {
  int32_t  width, height;
  int32_t  samples_per_pixel;
  int32_t  antialiasing;      /* 0: offsets fixed at 0.5 (RenderWorker.hs:98-99)          */
  int32_t  max_bounces;
  int32_t  rr_start_bounce;   /* 0 = off: the reference's behaviour ("Russian roulette removed", Config.hs:57) and what every parity
                                 run uses.  k > 0 (extension, BASELINE config 4): once a path has made k bounces each further bounce
                                 survives with probability q = clamp(max channel of the attenuation, 0.05, 1); a survivor's deeper value
                                 is weighted 1/q outside the reference's clamp, a terminated path adds nothing deeper.  Unbiased for the
                                 reference's estimator where its clamp(., 0, 10) is inactive. */
  double   exposure;          /* toneMap,      Trace.hs:96-101  */
  double   gamma;             /* gammaCorrect, Trace.hs:103-105 (used as the exponent) */
  uint64_t seed;
  int32_t  collect_stats;     /* 1: also count rays / nodes / prim tests (slower kernel variant) */
  int32_t  box_precision;     /* 0: default (the core's own SAH tree, FP32 conservative boxes, reference tie rule);
                                 1: the reference's tree with FP64 boxes, exact reference slab test, exactly its node visits;
                                 2: the reference's tree with FP32 conservative boxes.  All three give the same image. */
} rt_params;

/* Caller-allocated outputs, any of them may be NULL. */
typedef struct rt_output {
  double  *radiance;   /* 3*W*H  mean linear radiance  = averageColor, RenderWorker.hs:111-114 */
  double  *color;      /* 3*W*H  after toneMap + gammaCorrect = what pixelColor returns, RenderWorker.hs:76-82 */
  uint8_t *rgb8;       /* 3*W*H  truncate(c*255.999), RenderWorker.hs:108-109 (clamped to 0..255) */
  int32_t *prim_id;    /* W*H    prim_id of the first sample's primary hit, -1 = miss (or max_bounces = 0: nothing traced) */
} rt_output;

typedef struct rt_stats {
  double   ms_render;        /* device time of the render kernel(s), HIP events           */
  double   ms_total;         /* host wall time of the call                                 */
  uint64_t paths;            /* W*H*spp handled by this call                               */
  /* filled only when params.collect_stats != 0: */
  uint64_t rays_closest;     /* primary + bounce closest-hit BVH queries                   */
  uint64_t rays_shadow;      /* shadow (any-hit) BVH queries                               */
  uint64_t nodes_visited;    /* internal nodes fetched                                     */
  uint64_t tri_tests, sphere_tests, plane_tests;
  uint64_t shaded_hits;      /* material fetches                                           */
  /* scheduler census of the vote-scheduled kernel: wave-level trips through each body */
  uint64_t sched_node_steps, sched_prim_steps, sched_shade_steps;
  uint64_t top_node_hits;      /* node visits served by the LDS copy of the top BVH levels */
  uint64_t sched_live_lanes;   /* sum over all trips of the lanes that still had work */
  uint64_t cyc_vote, cyc_node, cyc_prim, cyc_shade, cyc_refill;   /* shader-clock ticks summed over waves, per body */
  uint64_t rays_shadow_culled; /* box_precision = 0 only: shadow queries of the reference NOT traced (and not in rays_shadow)
                                  because the surface faces away from the light: their term is max(0, n.l) = 0 either way */
  uint64_t rays_literal;       /* queries answered by the literal walk of the reference's tree: a plane was crossed beyond its leaf
                                  box's exit, where the reference's answer depends on its traversal history (BVH.hs:137-138) */
  /* census of the SHADE body's sub-blocks (hit, light term, shadow set-up, scatter, finish): lanes that ran each, and
   * wave-level trips in which at least one lane did; cyc_shade_detail: their time in experiment builds (-DRT_TIMERS):
   * loads, hit, light, set-up, scatter, finish, begin-query */
  uint64_t shade_lanes[5], shade_trips[5], cyc_shade_detail[7];
} rt_stats;

typedef struct rt_scene rt_scene;

/* Select the device this process renders on (one process per GPU).  device < 0 => current. */
int  rt_init(int device);
/* Select a GROUP of devices for a single-process host (the reference is one process that calls createPPM once,
 * app/Main.hs:53): scenes created afterwards are replicated on every device of the group and ONE blocking rt_render call
 * fans the frame out over them - tile (tx,ty) to device (tx+ty) mod n (see rt_tile_skew), one host thread and one stream per device, the
 * packed tiles brought to devices[0] as peer copies over xGMI - replacing the 24 green threads and the row TQueue of
 * app/Rendering/ImageGenerator.hs:37,64-94.  n_devices = 0 with devices = NULL selects every visible device.  Entries may
 * repeat ("virtual devices": how the tests exercise the fan-out on one GPU).  The frame is bit-identical for every n. */
int  rt_init_devices(int32_t n_devices, const int32_t *devices);
int  rt_device_count(void);   /* size of the current group (1 after rt_init) */
void rt_shutdown(void);
const char *rt_last_error(void);
int  rt_abi_version(void);
/* Measurement hooks (no counterpart in the reference).  rt_build_id: a digest of the kernel sources and compile-time knobs
 * this library was built from (Makefile) - committed counter records (profiles/pmc_traffic.json) carry it, so that bench.py
 * can tell a record taken on another kernel (stale) from one taken on the library it loaded.  rt_kernel_name: the
 * k_render_tiles instantiation a render of `scene` with `params` launches, spelt as rocprofv3 prints it (it depends on
 * box_precision, collect_stats, rr_start_bounce and on whether the scene has the core's own tree). */
const char *rt_build_id(void);
int  rt_kernel_name(rt_scene *scene, const rt_params *params, char *buf, int32_t cap);

/* Copies the descriptor's arrays to HBM in the kernels' layout; borrows nothing after return. */
int  rt_scene_create(const rt_scene_desc *desc, rt_scene **out);
void rt_scene_destroy(rt_scene *scene);

/* Diagnostic (no counterpart in the reference; needs no GPU): the render core's OWN acceleration structure, the
 * binned-SAH tree rt_scene_create builds over the scene's primitives and the kernels traverse when box_precision = 0
 * (the reference's tree still settles exact ties and bounds planes).  boxes: 6 doubles per primitive.
 * Outputs are caller-allocated for the worst case of 2*n_prims node indices: node_bounds[12*n_prims],
 * node_child[4*n_prims] (internal: left, right; leaf: -1 - first_slot, count; unused index: 0, 0), order[n_prims]
 * (slot -> primitive).  Root = node 0.  The result does not depend on n_threads. */
int  rt_fast_bvh_build(int32_t n_prims, const double *boxes, int32_t max_leaf, int32_t n_threads,
                       double *node_bounds, int32_t *node_child, int32_t *order, int32_t *depth);

/* One frame, host buffers in and out: the replacement of the worker fan-out. Blocking.  Renders on the device group the
 * scene was created for (one device after rt_init, n after rt_init_devices); stats->ms_render is the slowest device's
 * kernel time, the counters are summed over the devices. */
int  rt_render(rt_scene *scene, const rt_camera *cam, const rt_params *params,
               rt_output *out, rt_stats *stats);

/* ---- multi-GPU tile sharding (one process per GPU) ------------------------
 * The image is cut into 8x8-pixel tiles, tile (tx,ty) belongs to rank (tx+ty) mod world.
 * A rank's tiles, in increasing ty*tiles_x+tx order, are its "local tiles"; the packed
 * buffers below hold 64 pixel slots per local tile (slot = local_tile*64 + (py&7)*8 + (px&7)). */
int  rt_tile_count(int32_t width, int32_t height, int32_t rank, int32_t world);
/* (Experiment knob: the environment variable RT_TILE_SKEW = k, read ONCE per process, changes the map to (tx + k*ty) mod world
 * for every entry point of the library; rt_tile_skew returns the value in force, 1 by default.) */
int  rt_tile_skew(void);
/* tiles: out, rt_tile_count() entries of ty*tiles_x+tx */
int  rt_tile_list(int32_t width, int32_t height, int32_t rank, int32_t world, int32_t *tiles);

/* Renders this rank's tiles into DEVICE buffers (packed layout above), asynchronously on
 * `stream` (a hipStream_t, NULL = default stream).  d_color: 3*64*ntiles doubles (post
 * tone-map/gamma); d_radiance, d_prim_id optional (NULL).  The caller gathers the packed
 * buffers (RCCL) and un-tiles them with rt_untile / rt_untile_device.  stats->ms_render is
 * not filled (asynchronous); counters are filled (after a sync) only if collect_stats. */
int  rt_render_tiles_device(rt_scene *scene, const rt_camera *cam, const rt_params *params,
                            int32_t rank, int32_t world,
                            void *d_color, void *d_radiance, void *d_prim_id,
                            void *stream, rt_stats *stats);

/* The same share of the frame as the 8-BIT tiles only (packed, 3 bytes per slot): k_resolve applies showPixel's
 * quantisation (app/Rendering/RenderWorker.hs:108-109, per value) in the pass that averages the samples, so no colour
 * buffer and no quantise launch exist on this path, and 3 bytes per pixel are what the gather moves.  d_rgb8: 3*64*ntiles bytes. */
int  rt_render_tiles_rgb8_device(rt_scene *scene, const rt_camera *cam, const rt_params *params,
                                 int32_t rank, int32_t world, void *d_rgb8, void *stream, rt_stats *stats);

/* rt_render_tiles_device for callers that CANNOT ALLOCATE (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Same arguments, same results to the bit; but the launch
 * workspace (chunk counter, statistics, tile list, path records, per-path radiance) is a 128-MiB static device array
 * inside the library instead of buffers that grow on demand, and the call makes no hipMalloc / hipFree / event creation.
 * Limits: at most 262,144 tiles per rank (16.7 Mpixel shares); 64 * samples_per_pixel * 24 bytes (one tile's samples) must
 * fit 64 MiB (spp <= 43,690) - a frame whose samples exceed 64 MiB is rendered in several launches, which costs time, not
 * correctness; one workgroup's path records must fit 62 MiB (always true for BVH depth <= 60).  Beyond them:
 * RT_ERR_UNSUPPORTED.  Every call on a GPU shares that GPU's array: calls that overlap on one GPU race (the same holds
 * for two rt_render_tiles_device calls in flight on ONE rt_scene - at most one render in flight per scene, and successive
 * calls on a scene must be stream-ordered).  rt_kernel_times does not see these launches (creating an event allocates).
 * Under stream capture pass stats = NULL or params->collect_stats = 0: collecting the counters synchronises the stream and
 * copies them to the host (a blocking hipMemcpy), which is illegal while capturing; with collect_stats = 0 the call only
 * enqueues memsets and kernels on `stream` (tests/test_zz_no_alloc.py captures it into a graph and replays it). */
int  rt_render_tiles_device_noalloc(rt_scene *scene, const rt_camera *cam, const rt_params *params,
                                    int32_t rank, int32_t world,
                                    void *d_color, void *d_radiance, void *d_prim_id,
                                    void *stream, rt_stats *stats);

/* Measurement hook (no counterpart in the reference): sum of the durations, in ms, of the k_render_tiles launches
 * recorded on this scene since the previous call, taken with HIP events on the stream each launch went to, and
 * their number (at most 256 are kept).  Blocks until those launches have finished.  bench.py's roofline uses it.
 * For a scene on a device group: the largest per-device sum, and device 0's number of launches. */
int  rt_kernel_times(rt_scene *scene, double *sum_ms, int32_t *n_launches);

/* Scatters one rank's packed device buffer (n_comp doubles per slot) into a full
 * row-major device image of n_comp*W*H doubles, on `stream`. */
int  rt_untile_device(int32_t width, int32_t height, int32_t rank, int32_t world,
                      const void *d_packed, void *d_image, int32_t n_comp, void *stream);

/* The same for slots of n_bytes BYTES each (no alignment assumed): used to gather the 8-bit frame, i.e. packed colours
 * quantised per rank with rt_quantize_device (3 bytes per pixel over xGMI instead of 24). */
int  rt_untile_bytes_device(int32_t width, int32_t height, int32_t rank, int32_t world,
                            const void *d_packed, void *d_image, int32_t n_bytes, void *stream);

/* The gathered packed buffers of ALL ranks - rank r's slots start at d_packed_all + r * rank_stride_bytes, n_bytes per slot -
 * scattered into the row-major device image of n_bytes*W*H bytes in ONE launch (rank 0 after the gather: one launch instead
 * of `world`). */
int  rt_untile_all_bytes_device(int32_t width, int32_t height, int32_t world, const void *d_packed_all, int64_t rank_stride_bytes,
                                void *d_image, int32_t n_bytes, void *stream);

/* Device image of 3*W*H doubles (color) -> 3*W*H uint8 (showPixel quantisation), on `stream`. */
int  rt_quantize_device(const void *d_color, void *d_rgb8, int64_t n_values, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RTCORE_H */
