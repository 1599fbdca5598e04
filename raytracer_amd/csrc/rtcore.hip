// rtcore.hip — C ABI (include/rtcore.h) of the MI355X render core: scene re-layout into HBM,
// kernel launches, tile sharding helpers.  gfx950 only; there is NO CPU fallback: without a HIP
// device every entry point returns RT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef RT_TOP_NODES_MAX
#define RT_TOP_NODES_MAX 1023
#endif
#ifndef RT_CHUNKS_PER_WAVE
#define RT_CHUNKS_PER_WAVE 25
#endif
#include "rt_kernels.h"
#include "rt_fastbvh.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(RT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));  \
  } while (0)

// One entry per HIP device this process has initialised (rt_init / rt_init_devices).
constexpr int RT_MAX_DEVICES = 64;
struct DevCtx {
  bool ready = false;
  int n_cu = 0;
  unsigned char *scratch = nullptr;     // this device's copy of the static scratch array below (hipGetSymbolAddress)
};
DevCtx g_dev[RT_MAX_DEVICES];
std::mutex g_init_mu;
// The devices one rt_render call fans out over (rt_init_devices); entries may repeat ("virtual devices": tests on one GPU).
std::vector<int> g_group;

}  // namespace

// ---- scratch of the entry point that allocates nothing (rt_render_tiles_device_noalloc) -------------------------------
// A file-scope device array: every GPU the library runs on holds its own copy, at its own address, until the process
// ends.  In host code the NAME `scratch` is only a handle for hipGetSymbolAddress - never a device address; the address
// is resolved once per device in init_device() and kept in g_dev[device].scratch, and that pointer is what kernels,
// hipMemsetAsync and copies receive.  Layout (all offsets multiples of 256):
//   [0, 4 KiB)            chunk counter (4 B) at 0, STAT_COUNT counters at 256
//   [4 KiB, +1 MiB)       this rank's tile list: at most RT_SCRATCH_MAX_TILES entries
//   [.., 64 MiB)          path records (cold rows + stack overflow rows): as many resident waves as fit
//   [64 MiB, 128 MiB)     per-path radiance of one launch: the frame is rendered in as many launches as that takes
#define RT_SCRATCH_BYTES ((size_t)128 << 20)
#define RT_SCRATCH_TILES_OFF ((size_t)4096)
#define RT_SCRATCH_MAX_TILES ((size_t)262144)
#define RT_SCRATCH_REC_OFF (RT_SCRATCH_TILES_OFF + RT_SCRATCH_MAX_TILES * 4)
#define RT_SCRATCH_SAMPLES_OFF ((size_t)64 << 20)
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];

namespace {

int init_device(int device) {            // with `device` current; idempotent
  std::lock_guard<std::mutex> g(g_init_mu);
  DevCtx &c = g_dev[device];
  if (c.ready) return RT_OK;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  c.n_cu = prop.multiProcessorCount;
  void *p = nullptr;
  HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));     // this device's copy of the array
  if (!p) return fail(RT_ERR_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
  c.scratch = (unsigned char *)p;
  c.ready = true;
  return RT_OK;
}

// Entry points that can be the first call of a process initialise the calling thread's current device - but ONLY when nothing
// has been initialised yet: a group chosen by rt_init / rt_init_devices is never replaced implicitly (a call from a fresh OS
// thread, whose current device is 0 whatever the group is, must not turn the group [2, 3] into [0]).
std::vector<int> group_snapshot() {
  std::lock_guard<std::mutex> g(g_init_mu);
  return g_group;
}
int ensure_init() {
  if (!group_snapshot().empty()) return RT_OK;
  return rt_init(-1);
}

float f32_down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; }
float f32_up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; }

// f(begin, end) over [0, n) in contiguous pieces on up to n_threads host threads (set-up loops over millions of primitives)
template <typename F> void parallel_for(size_t n, int n_threads, F f) {
  const size_t pieces = std::max<size_t>(1, std::min<size_t>((size_t)std::max(1, n_threads), n / 16384));
  if (pieces == 1) { f((size_t)0, n); return; }
  std::vector<std::thread> th;
  for (size_t k = 0; k < pieces; ++k) th.emplace_back([=]() { f(n * k / pieces, n * (k + 1) / pieces); });
  for (auto &t : th) t.join();
}
int host_threads() {
  int nt = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  if (const char *env = getenv("RT_BUILD_THREADS")) nt = std::max(1, atoi(env));
  return nt;
}

template <typename T> int dev_upload(T **dst, const T *src, size_t n) {
  *dst = nullptr;
  size_t bytes = std::max<size_t>(n, 1) * sizeof(T);
  if (hipMalloc((void **)dst, bytes) != hipSuccess) return fail(RT_ERR_OOM, "hipMalloc of %zu bytes failed", bytes);
  if (n) HIP_TRY(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
  return RT_OK;
}

}  // namespace

// Tile (tx, ty) belongs to rank (tx + skew * ty) mod world.  skew = 1 unless RT_TILE_SKEW says otherwise (experiment knob of
// tools/rank_timing.py; raytracer_amd/dist.py reads the same variable): read once, so every list of a process agrees.
static int tile_skew() {
  static const int v = []() { const char *e = getenv("RT_TILE_SKEW"); const int k = e ? atoi(e) : 1; return k > 0 ? k : 1; }();
  return v;
}
static inline int tiles_x_of(int W) { return (W + 7) / 8; }
static inline int tiles_y_of(int H) { return (H + 7) / 8; }

// What a launch needs beyond the scene, its inputs and its outputs.  Either owned by a Replica (hipMalloc'ed, grows on
// demand, never inside a steady-state loop) or carved out of the static scratch array (fixed: rt_render_tiles_device_noalloc).
struct Workspace {
  uint32_t *counter = nullptr;          // chunk counter of the persistent kernel
  unsigned long long *stats = nullptr;  // STAT_COUNT counters
  int32_t *tiles = nullptr;             // this rank's tile list (k_make_tiles)
  size_t tiles_cap = 0;                 // entries
  double *fscratch = nullptr;           // per resident wave: the lanes' path records (rt_render_kernel.h)
  size_t fscratch_cap = 0;              // bytes
  double *samples = nullptr;            // radiance of every path of one launch, [item][3]
  size_t samples_cap = 0;               // bytes
  bool fixed = false;                   // true: capacities never change and nothing is ever allocated
  int tl_w = -1, tl_h = -1, tl_rank = -1, tl_world = -1, tl_n = 0;   // what `tiles` currently holds
};

// The scene in the kernels' layout, on the host: built once per rt_scene_create, uploaded to every device of the group.
struct HostScene {
  std::vector<DNode<float>> n32, fn32;
  std::vector<DNode<double>> n64;
  std::vector<DPrim> prims, fprims;
  std::vector<int32_t> pids, ref_anc, ref_depth, leaf_anc;
  std::vector<DMaterial> mats;
  std::vector<DLight> lights;
  std::vector<double> plane_boxes, tri_normals;
  std::vector<double> sky_u_tab, sky_u_q, sky_v_tab;   // texel boundaries of the sky lookup (sky_texel, rt_kernels.h)
  const void *sky = nullptr;
  size_t sky_bytes = 0;
  bool has_fast = false;
  bool inverted_boxes = false;          // a sphere of negative radius: only the exact FP64 slab test reproduces the reference
  int n_internal = 0, depth = 0, f_internal = 0, f_depth = 0, f_root_ref = 0;
  KScene ks{};                          // the non-pointer fields
};

// One device's copy of the scene + its launch workspace + cached outputs.
struct Replica {
  int device = 0;
  // HBM-resident scene
  DNode<float> *d_nodes32 = nullptr;
  DNode<double> *d_nodes64 = nullptr;   // the reference tree's FP64 boxes: box_precision = 1, the tie rule of the fast tree, literal_hit
  DPrim *d_prims = nullptr;
  // the render core's own tree (rt_fastbvh.h); absent when the reference root is a leaf
  bool has_fast = false;
  DNode<float> *d_fnodes32 = nullptr;
  DPrim *d_fprims = nullptr;
  int32_t *d_ref_anc = nullptr, *d_ref_depth = nullptr, *d_leaf_anc = nullptr;
  int f_internal = 0, f_depth = 0, f_root_ref = 0;
  int32_t *d_prim_ids = nullptr;
  DMaterial *d_mats = nullptr;
  DLight *d_lights = nullptr;
  double *d_plane_boxes = nullptr;
  double *d_tri_normals = nullptr;
  void *d_sky = nullptr;
  double *d_sky_u_tab = nullptr, *d_sky_u_q = nullptr;
  KScene ks{};
  int n_internal = 0, depth = 0;
  bool has_boxed_planes = false;        // some plane sits in a non-root leaf: its visibility can depend on the reference's traversal (literal_hit)
  bool inverted_boxes = false;          // every box_precision runs the exact FP64 path
  Workspace ws;                         // owned, growable
  // cached outputs of rt_render: full frame (single device, or device 0 of a group) ...
  size_t cap_px = 0;
  double *d_color = nullptr, *d_radiance = nullptr;
  int32_t *d_primid = nullptr;
  uint8_t *d_rgb8 = nullptr;
  // ... and, in a device group, this replica's packed tiles (on its device) and where they land on device 0
  size_t cap_slots = 0;
  double *p_color = nullptr, *p_radiance = nullptr;
  int32_t *p_primid = nullptr;
  uint8_t *p_rgb8 = nullptr;
  double *g_color = nullptr, *g_radiance = nullptr;     // where they land on the group's device 0: slices of rep[0]'s gather areas
  int32_t *g_primid = nullptr;                          // (replica 0 renders straight into its slices: its p_ pointers alias them)
  uint8_t *g_rgb8 = nullptr;
  // replica 0 only: the gather areas, n replicas x cap_slots slots each, contiguous so that ONE k_untile_all launch scatters them
  double *G_color = nullptr, *G_radiance = nullptr;
  int32_t *G_primid = nullptr;
  uint8_t *G_rgb8 = nullptr;
  int G_n = 0;
  hipStream_t stream = nullptr;         // this replica's stream in a device group
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // HIP event pairs around the k_render_tiles launches since the last rt_kernel_times call (measurement hook)
  static constexpr int KT_RING = 256;
  std::vector<hipEvent_t> kt_events;    // 2 per launch, created on first use
  int kt_pending = 0;
};

struct rt_scene {
  std::vector<Replica *> rep;           // one per device of the group the scene was created for (rep[0]: where frames are assembled)
};

static void replica_destroy(Replica *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  (void)hipFree(s->d_nodes32); (void)hipFree(s->d_nodes64); (void)hipFree(s->d_prims); (void)hipFree(s->d_prim_ids);
  (void)hipFree(s->d_fnodes32); (void)hipFree(s->d_fprims); (void)hipFree(s->d_ref_anc); (void)hipFree(s->d_ref_depth); (void)hipFree(s->d_leaf_anc);
  (void)hipFree(s->d_mats); (void)hipFree(s->d_lights); (void)hipFree(s->d_plane_boxes); (void)hipFree(s->d_tri_normals); (void)hipFree(s->d_sky); (void)hipFree(s->d_sky_u_tab); (void)hipFree(s->d_sky_u_q);
  (void)hipFree(s->ws.counter); (void)hipFree(s->ws.stats); (void)hipFree(s->ws.fscratch); (void)hipFree(s->ws.samples); (void)hipFree(s->ws.tiles);
  (void)hipFree(s->d_color); (void)hipFree(s->d_radiance); (void)hipFree(s->d_primid); (void)hipFree(s->d_rgb8);
  if (s->p_color != s->g_color) { (void)hipFree(s->p_color); (void)hipFree(s->p_radiance); (void)hipFree(s->p_primid); (void)hipFree(s->p_rgb8); }
  (void)hipFree(s->G_color); (void)hipFree(s->G_radiance); (void)hipFree(s->G_primid); (void)hipFree(s->G_rgb8);
  for (hipEvent_t e : s->kt_events) (void)hipEventDestroy(e);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  if (s->stream) (void)hipStreamDestroy(s->stream);
  delete s;
}

// Validates the descriptor and lays the scene out for the kernels (host memory only).
static int prepare_host_scene(const rt_scene_desc *d, HostScene &hs) {
  int rc;
  if (d->n_nodes <= 0 || d->n_prims <= 0) return fail(RT_ERR_EMPTY_SCENE, "empty scene: Cannot compute bounding box of empty list");
  if (!d->node_bounds || !d->node_child || !d->prim_kind || !d->prim_data || !d->prim_material)
    return fail(RT_ERR_INVALID, "scene descriptor has null arrays");
  if (d->n_materials < 0 || d->n_lights < 0 || (d->n_materials && !d->materials) || (d->n_lights && !d->lights))
    return fail(RT_ERR_INVALID, "bad material/light table");
  if (d->sky_format != RT_SKY_NONE && (d->sky_width <= 0 || d->sky_height <= 0 || !d->sky_pixels))
    return fail(RT_ERR_INVALID, "bad sky texture");
  if (d->sky_format < RT_SKY_NONE || d->sky_format > RT_SKY_RGBF) return fail(RT_ERR_INVALID, "bad sky format");
  if (d->bg_kind != RT_BG_GRADIENT && d->bg_kind != RT_BG_SOLID) return fail(RT_ERR_INVALID, "unknown background kind");
  const int nn = d->n_nodes, np = d->n_prims;
  const bool dbg_t = getenv("RT_DEBUG") != nullptr;          // RT_DEBUG: where the set-up time goes
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "rtcore: scene_create %-28s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  // ---- validate the tree (every kernel index is checked here, on the host) and order the internal nodes:
  //      the top levels breadth-first (they are the ones every ray touches), the subtrees below depth-first.
  std::vector<int> level(nn, -1), dev_index(nn, -1), order;
  std::vector<char> seen(nn, 0);
  auto is_leaf = [&](int i) { return d->node_child[2 * i] < 0; };
  auto check_node = [&](int i) -> int {
    if (i < 0 || i >= nn) return fail(RT_ERR_INVALID, "node index %d out of range", i);
    if (seen[i]) return fail(RT_ERR_INVALID, "node %d reachable twice: not a tree", i);
    seen[i] = 1;
    if (is_leaf(i)) {
      long first = -1L - d->node_child[2 * i], cnt = d->node_child[2 * i + 1];
      if (first < 0 || cnt < 1 || first + cnt > np) return fail(RT_ERR_INVALID, "leaf %d has bad primitive range", i);
    } else if (d->node_child[2 * i] == 0 || d->node_child[2 * i + 1] <= 0) {
      return fail(RT_ERR_INVALID, "node %d has a bad child index", i);
    }
    return RT_OK;
  };
  const int TOP_LEVELS = 10;
  int depth = 0;
  if ((rc = check_node(0))) return rc;
  level[0] = 0;
  std::vector<int> frontier{0}, deep_roots;
  for (int lv = 0; !frontier.empty(); ++lv) {
    std::vector<int> next;
    for (int i : frontier) {
      depth = std::max(depth, lv);
      if (is_leaf(i)) continue;
      if (lv >= TOP_LEVELS) { deep_roots.push_back(i); continue; }
      dev_index[i] = (int)order.size(); order.push_back(i);
      for (int k = 0; k < 2; ++k) {
        int c = d->node_child[2 * i + k];
        if ((rc = check_node(c))) return rc;
        level[c] = lv + 1; next.push_back(c);
      }
    }
    frontier.swap(next);
  }
  for (int r : deep_roots) {   // depth-first (pre-order) below the top levels
    std::vector<int> stk{r};
    while (!stk.empty()) {
      int i = stk.back(); stk.pop_back();
      depth = std::max(depth, level[i]);
      if (is_leaf(i)) continue;
      dev_index[i] = (int)order.size(); order.push_back(i);
      for (int k = 1; k >= 0; --k) {
        int c = d->node_child[2 * i + k];
        if ((rc = check_node(c))) return rc;
        level[c] = level[i] + 1; stk.push_back(c);
      }
    }
  }
  if (depth > 60) return fail(RT_ERR_UNSUPPORTED, "BVH depth %d exceeds the traversal stack limit (60)", depth);

  lap("validate + order ref nodes");
  HostScene *s = &hs;
  s->depth = depth;
  s->n_internal = (int)order.size();

  // ---- materials: dense table, index 0 = defaultMaterial (Rendering/Material.hs:25-35)
  std::vector<DMaterial> &mats = hs.mats;
  mats.assign(1 + d->n_materials, DMaterial());
  auto set_mat = [](DMaterial &m, const rt_material *src) {
    std::memset(&m, 0, sizeof m);
    if (!src) { m.diffuse[0] = m.diffuse[1] = m.diffuse[2] = 1.0; return; }
    for (int k = 0; k < 3; ++k) { m.diffuse[k] = src->diffuse[k]; m.emission[k] = (src->flags & RT_MAT_HAS_EMISSION) ? src->emission[k] : 0.0; }
    const bool trans1 = (src->flags & RT_MAT_HAS_TRANSMISSION) && src->transmission == 1.0;
    m.ior = src->ior;
    if (trans1) m.flags |= DMAT_NO_ATTEN;
    if (trans1 && (src->flags & RT_MAT_HAS_IOR)) m.flags |= DMAT_GLASS_DIR;
    if ((src->flags & RT_MAT_HAS_SHININESS) && src->shininess > 100) m.flags |= DMAT_METAL;
  };
  set_mat(mats[0], nullptr);
  for (int i = 0; i < d->n_materials; ++i) set_mat(mats[1 + i], &d->materials[i]);
  auto mat_index = [&](int id) {   // M.lookup: the LAST entry with that key wins (M.fromList)
    for (int i = d->n_materials - 1; i >= 0; --i) if (d->materials[i].id == id) return 1 + i;
    return 0;
  };
  // material id -> dense index, resolved once per distinct id (a scene has a handful)
  std::vector<int> mat_ids, mat_ixs;
  for (int i = 0; i < np; ++i) {
    const int mid = d->prim_material[i];
    if (std::find(mat_ids.begin(), mat_ids.end(), mid) == mat_ids.end()) { mat_ids.push_back(mid); mat_ixs.push_back(mat_index(mid)); }
  }
  const int n_host_threads = host_threads();

  // ---- primitives (leaf order is kept: it decides ties) + the FP64 leaf boxes planes need
  std::vector<DPrim> &prims = hs.prims;
  std::vector<int32_t> &pids = hs.pids;
  std::vector<double> &plane_boxes = hs.plane_boxes;
  prims.assign(np, DPrim()); pids.assign(np, 0); plane_boxes.clear();
  // Per triangle: the face normal the reference recomputes at every hit, normalize(e1 x e2) (Triangle.hs:51), and the
  // re-normalisation computeLighting applies to it (Light.hs:19).  Same operations in the same order as the device
  // helpers cross3 / dot3 / normalize3 (rt_device.h), IEEE sqrt and division, no contraction: bit-identical values.
  std::vector<double> &tri_normals = hs.tri_normals;
  tri_normals.assign((size_t)np * 8, 0.0);   // 64-byte shading records: n, normalize(n), (material, kind), pad
  auto normalize_like_device = [](const double v[3], double out[3]) {
    const double s = 1.0 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    out[0] = s * v[0]; out[1] = s * v[1]; out[2] = s * v[2];
  };
  double maxabs = 0;
  std::atomic<int> bad_prim(-1);
  parallel_for((size_t)np, n_host_threads, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; ++i) {
      const double *src = d->prim_data + 9 * i;
      DPrim &P = prims[i];
      std::memset(&P, 0, sizeof P);
      const int kind = d->prim_kind[i];
      if (kind < RT_PRIM_TRIANGLE || kind > RT_PRIM_PLANE) { bad_prim.store((int)i); return; }
      if (kind == RT_PRIM_TRIANGLE) {
        for (int k = 0; k < 3; ++k) {
          P.d[k] = src[k];
          P.d[3 + k] = src[3 + k] - src[k];   // e1 = p1 - p0, Triangle.hs:32
          P.d[6 + k] = src[6 + k] - src[k];   // e2 = p2 - p0, Triangle.hs:33
        }
        const double *a = P.d + 3, *b = P.d + 6;
        const double c[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        normalize_like_device(c, &tri_normals[8 * i]);
        normalize_like_device(&tri_normals[8 * i], &tri_normals[8 * i + 3]);
      } else {
        for (int k = 0; k < 9; ++k) P.d[k] = src[k];
        if (kind == RT_PRIM_PLANE) P.d[6] = -1.0;
      }
      P.kind = (uint32_t)kind;
      P.mat = mat_ixs[std::find(mat_ids.begin(), mat_ids.end(), d->prim_material[i]) - mat_ids.begin()];
      pids[i] = d->prim_id ? d->prim_id[i] : (int32_t)i;
    }
  });
  if (bad_prim.load() >= 0) { const int i = bad_prim.load(); return fail(RT_ERR_INVALID, "primitive %d has unknown kind %d", i, d->prim_kind[i]); }
  lap("primitives + normals");
  // leaves: mark the last primitive, give planes their leaf box
  std::vector<char> in_leaf(np, 0);
  for (int i = 0; i < nn; ++i) {
    if (!seen[i] || !is_leaf(i)) continue;
    long first = -1L - d->node_child[2 * i], cnt = d->node_child[2 * i + 1];
    prims[first + cnt - 1].kind |= DPRIM_LAST;
    for (long k = first; k < first + cnt; ++k) {
      if (in_leaf[k]) { return fail(RT_ERR_INVALID, "primitive %ld belongs to two leaves", k); }
      in_leaf[k] = 1;
      const uint32_t kd = prims[k].kind & DPRIM_KIND_MASK;
      if (kd == RT_PRIM_SPHERE) prims[k].d[6] = -1.0;
      // planes (their box does not contain them) and spheres (they touch their unpadded box; radius <= 0: inside-out / point boxes)
      // are tested against the interval their leaf RECEIVES; the root's own box is never tested (BVH.hs:123)
      if ((kd == RT_PRIM_PLANE || kd == RT_PRIM_SPHERE) && i != 0) {
        prims[k].d[6] = (double)(plane_boxes.size() / 6);
        for (int q = 0; q < 6; ++q) plane_boxes.push_back(d->node_bounds[6 * (size_t)i + q]);
      }
    }
  }
  for (int i = 0; i < np; ++i)
    if (!in_leaf[i]) { return fail(RT_ERR_INVALID, "primitive %d belongs to no leaf of the tree", i); }
  for (int i = 0; i < np; ++i) {                              // the (material, kind) pair of the shading records
    const int32_t pr[2] = {prims[i].mat, (int32_t)prims[i].kind};
    std::memcpy(&tri_normals[8 * (size_t)i + 6], pr, sizeof pr);
  }
  for (int q = 0; q < 6; ++q) maxabs = std::max(maxabs, std::fabs(d->node_bounds[q]));

  // ---- internal nodes: children's boxes, FP64 as given and FP32 rounded outward
  const int ni = std::max(1, s->n_internal);
  std::vector<DNode<float>> &n32 = hs.n32;
  std::vector<DNode<double>> &n64 = hs.n64;
  n32.resize(ni); n64.resize(ni);
  std::memset(n32.data(), 0, ni * sizeof(DNode<float>));
  std::memset(n64.data(), 0, ni * sizeof(DNode<double>));
  // where each internal node and each primitive hangs in the reference tree (the fast tree's tie rule climbs these)
  std::vector<int32_t> &ref_anc = hs.ref_anc, &ref_depth = hs.ref_depth, &leaf_anc = hs.leaf_anc;
  ref_anc.assign(ni, -1); ref_depth.assign(ni, 0); leaf_anc.assign(np, 0);
  auto ref_of = [&](int child) -> int32_t { return is_leaf(child) ? d->node_child[2 * child] : dev_index[child]; };   // leaf: -1-first == ~first
  for (int k = 0; k < s->n_internal; ++k) {
    int i = order[k];
    for (int c = 0; c < 2; ++c) {
      int child = d->node_child[2 * i + c];
      const double *b = d->node_bounds + 6 * (size_t)child;
      for (int q = 0; q < 3; ++q) {
        n64[k].box[6 * c + q] = b[q];
        n64[k].box[6 * c + 3 + q] = b[3 + q];
        n32[k].box[6 * c + q] = f32_down(b[q]);
        n32[k].box[6 * c + 3 + q] = f32_up(b[3 + q]);
      }
      n64[k].ref[c] = n32[k].ref[c] = ref_of(child);
      if (is_leaf(child)) {
        long first = -1L - d->node_child[2 * child], cnt = d->node_child[2 * child + 1];
        for (long q = first; q < first + cnt; ++q) leaf_anc[q] = (k << 1) | c;
      } else {
        ref_anc[dev_index[child]] = (k << 1) | c;     // parents precede their children in `order`
        ref_depth[dev_index[child]] = ref_depth[k] + 1;
      }
    }
  }

  lap("leaves + ref node arrays");
  // ---- the render core's own tree over the same primitives (rt_fastbvh.h).  Boxes: triangles as Triangle.hs:24-28
  //      (vertex extent padded by 1e-8), spheres as Sphere.hs:20-22, planes by the box of their reference leaf -
  //      the only region where the reference can see them (Plane.hs:20).
  FastBvh fb;
  std::vector<DNode<float>> &fn32 = hs.fn32;
  std::vector<DPrim> &fprims = hs.fprims;
  fn32.resize(1); fprims.clear();
  std::memset(fn32.data(), 0, sizeof(DNode<float>));
  bool fast_ok = !is_leaf(0);
  if (fast_ok) {
    std::vector<double> pbox((size_t)np * 6);
    std::atomic<bool> finite(true), inverted(false);
    parallel_for((size_t)np, n_host_threads, [&](size_t i0, size_t i1) {
      for (size_t i = i0; i < i1; ++i) {
        double *b = &pbox[6 * i];
        const uint32_t kind = prims[i].kind & DPRIM_KIND_MASK;
        if (kind == RT_PRIM_TRIANGLE) {
          const double *v = d->prim_data + 9 * i;
          for (int q = 0; q < 3; ++q) {
            b[q] = std::min(v[q], std::min(v[3 + q], v[6 + q])) - 1e-8;
            b[3 + q] = std::max(v[q], std::max(v[3 + q], v[6 + q])) + 1e-8;
          }
        } else if (kind == RT_PRIM_SPHERE) {
          // a negative radius gives the reference an inside-out box (Sphere.hs:20-22: c - r > c + r) that its slab test and
          // unions treat in their own way: such scenes keep to the reference's own tree and its exact FP64 slab test
          // (the outward rounding and the origin margin of the FP32 boxes assume lo <= hi)
          if (prims[i].d[3] < 0.0) { finite.store(false); inverted.store(true); }
          const double r = std::fabs(prims[i].d[3]);
          for (int q = 0; q < 3; ++q) { b[q] = prims[i].d[q] - r; b[3 + q] = prims[i].d[q] + r; }
        } else {
          const double *lb = &plane_boxes[6 * (size_t)prims[i].d[6]];
          for (int q = 0; q < 6; ++q) b[q] = lb[q];
        }
        for (int q = 0; q < 6; ++q) if (!std::isfinite(b[q])) finite.store(false);
      }
    });
    fast_ok = finite.load();
    hs.inverted_boxes = inverted.load();
    if (fast_ok) {
      const int nt = n_host_threads;
      int max_leaf = 4;
      double cost_isect = 1.2;
      if (const char *env = getenv("RT_FAST_LEAF")) max_leaf = std::max(1, atoi(env));        // experiment knobs
      if (const char *env = getenv("RT_FAST_CI")) cost_isect = atof(env);
      lap("primitive boxes");
      build_fast_bvh(np, pbox.data(), max_leaf, cost_isect, nt, fb);
      lap("fast tree build");
      if (fb.depth > 60) fast_ok = false;
    }
  }
  if (fast_ok) {
    // internal nodes: the top levels breadth-first (LDS cache), the subtrees below depth-first, as for the reference tree
    auto f_leaf = [&](int i) { return fb.child[2 * (size_t)i] < 0; };
    std::vector<int> f_order, f_index(fb.child.size() / 2, -1), fr_{0}, f_deep;
    for (int lv = 0; !fr_.empty(); ++lv) {
      std::vector<int> next;
      for (int i : fr_) {
        if (f_leaf(i)) continue;
        if (lv >= TOP_LEVELS) { f_deep.push_back(i); continue; }
        f_index[i] = (int)f_order.size(); f_order.push_back(i);
        next.push_back(fb.child[2 * (size_t)i]); next.push_back(fb.child[2 * (size_t)i + 1]);
      }
      fr_.swap(next);
    }
    for (int r : f_deep) {
      std::vector<int> stk{r};
      while (!stk.empty()) {
        int i = stk.back(); stk.pop_back();
        if (f_leaf(i)) continue;
        f_index[i] = (int)f_order.size(); f_order.push_back(i);
        stk.push_back(fb.child[2 * (size_t)i + 1]); stk.push_back(fb.child[2 * (size_t)i]);
      }
    }
    s->f_internal = (int)f_order.size();
    s->f_depth = fb.depth;
    s->f_root_ref = f_leaf(0) ? fb.child[0] : 0;
    fn32.assign(std::max(1, s->f_internal), DNode<float>());
    std::memset(fn32.data(), 0, fn32.size() * sizeof(DNode<float>));
    for (int k = 0; k < s->f_internal; ++k) {
      const int i = f_order[k];
      for (int c = 0; c < 2; ++c) {
        const int child = fb.child[2 * (size_t)i + c];
        const double *b = &fb.bounds[6 * (size_t)child];
        for (int q = 0; q < 3; ++q) { fn32[k].box[6 * c + q] = f32_down(b[q]); fn32[k].box[6 * c + 3 + q] = f32_up(b[3 + q]); }
        fn32[k].ref[c] = f_leaf(child) ? fb.child[2 * (size_t)child] : f_index[child];
      }
    }
    // primitives in this tree's leaf order; the material slot carries the index in the reference's order instead
    fprims.resize(np);
    parallel_for((size_t)np, n_host_threads, [&](size_t q0, size_t q1) {
      for (size_t q = q0; q < q1; ++q) { fprims[q] = prims[fb.order[q]]; fprims[q].mat = fb.order[q]; fprims[q].kind &= ~DPRIM_LAST; }
    });
    for (size_t i = 0; i < fb.child.size() / 2; ++i)
      if (f_leaf((int)i)) fprims[(size_t)(-1 - fb.child[2 * i]) + fb.child[2 * i + 1] - 1].kind |= DPRIM_LAST;
  }
  s->has_fast = fast_ok;
  lap("fast tree order + prims");

  std::vector<DLight> &lights = hs.lights;
  lights.resize(std::max(1, d->n_lights));
  std::memset(lights.data(), 0, lights.size() * sizeof(DLight));
  for (int i = 0; i < d->n_lights; ++i) {
    if (d->lights[i].kind != RT_LIGHT_POINT && d->lights[i].kind != RT_LIGHT_DIRECTIONAL) { return fail(RT_ERR_INVALID, "light %d has unknown kind", i); }
    for (int k = 0; k < 3; ++k) { lights[i].vec[k] = d->lights[i].vec[k]; lights[i].intensity[k] = d->lights[i].intensity[k]; }
    lights[i].kind = d->lights[i].kind;
  }


  hs.sky = d->sky_pixels;
  if (d->sky_format != RT_SKY_NONE) {
    skytab::build(d->sky_width, d->sky_height, SKY_TAB_PAD, hs.sky_u_tab, hs.sky_u_q, hs.sky_v_tab);   // rt_sky_tables.h
  }
  hs.sky_bytes = d->sky_format == RT_SKY_NONE ? 0 : (size_t)d->sky_width * d->sky_height * 3 * (d->sky_format == RT_SKY_RGB8 ? 1 : 4);
  KScene &k = hs.ks;
  k.root_ref = is_leaf(0) ? d->node_child[0] : 0;
  k.ref_root_ref = k.root_ref;
  k.n_lights = d->n_lights;
  k.bg_kind = d->bg_kind; k.sky_format = d->sky_format; k.sky_w = d->sky_width; k.sky_h = d->sky_height;
  for (int q = 0; q < 3; ++q) { k.bg1[q] = d->bg_color1[q]; k.bg2[q] = d->bg_color2[q]; }
  k.rscene = f32_up(maxabs);
  lap("lights + scalars");
  return RT_OK;
}

// Copies a prepared scene to the CURRENT device.
static int upload_replica(const HostScene &hs, int device, Replica **out) {
  int rc;
  auto *s = new Replica();
  s->device = device;
  s->depth = hs.depth; s->n_internal = hs.n_internal;
  s->has_fast = hs.has_fast; s->f_internal = hs.f_internal; s->f_depth = hs.f_depth; s->f_root_ref = hs.f_root_ref;
  s->has_boxed_planes = !hs.plane_boxes.empty();
  s->inverted_boxes = hs.inverted_boxes;
#define UP(call) do { if ((rc = (call))) { replica_destroy(s); return rc; } } while (0)
  UP(dev_upload(&s->d_nodes32, hs.n32.data(), hs.n32.size()));
  UP(dev_upload(&s->d_nodes64, hs.n64.data(), hs.n64.size()));
  if (hs.has_fast) {
    UP(dev_upload(&s->d_fnodes32, hs.fn32.data(), hs.fn32.size()));
    UP(dev_upload(&s->d_fprims, hs.fprims.data(), hs.fprims.size()));
  }
  UP(dev_upload(&s->d_ref_anc, hs.ref_anc.data(), hs.ref_anc.size()));       // the tie rule: box_precision 0 and 2
  UP(dev_upload(&s->d_ref_depth, hs.ref_depth.data(), hs.ref_depth.size()));
  UP(dev_upload(&s->d_leaf_anc, hs.leaf_anc.data(), hs.leaf_anc.size()));
  UP(dev_upload(&s->d_prims, hs.prims.data(), hs.prims.size()));
  UP(dev_upload(&s->d_prim_ids, hs.pids.data(), hs.pids.size()));
  UP(dev_upload(&s->d_mats, hs.mats.data(), hs.mats.size()));
  UP(dev_upload(&s->d_lights, hs.lights.data(), hs.lights.size()));
  UP(dev_upload(&s->d_plane_boxes, hs.plane_boxes.data(), hs.plane_boxes.size()));
  UP(dev_upload(&s->d_tri_normals, hs.tri_normals.data(), hs.tri_normals.size()));
  if (hs.sky_bytes) {
    uint8_t *p = nullptr;
    UP(dev_upload(&p, (const uint8_t *)hs.sky, hs.sky_bytes));
    s->d_sky = p;
    UP(dev_upload(&s->d_sky_u_tab, hs.sky_u_tab.data(), hs.sky_u_tab.size()));
    {                                                  // one table: (w, h), Q_k[3] ..., T_k ...: sky_texel_exact takes one pointer
      std::vector<double> t{(double)hs.ks.sky_w, (double)hs.ks.sky_h};
      t.insert(t.end(), hs.sky_u_q.begin(), hs.sky_u_q.end());
      t.insert(t.end(), hs.sky_v_tab.begin(), hs.sky_v_tab.end());
      UP(dev_upload(&s->d_sky_u_q, t.data(), t.size()));
    }
  }
  uint32_t zero = 0;
  UP(dev_upload(&s->ws.counter, &zero, 1));
  unsigned long long zs[STAT_COUNT] = {0};
  UP(dev_upload(&s->ws.stats, zs, STAT_COUNT));
#undef UP
  if (hipEventCreate(&s->ev0) != hipSuccess || hipEventCreate(&s->ev1) != hipSuccess) { replica_destroy(s); return fail(RT_ERR_HIP, "hipEventCreate failed"); }
  KScene &k = s->ks;
  k = hs.ks;
  k.nodes = s->d_nodes32; k.prims = s->d_prims; k.tprims = s->d_prims;
  k.ref_nodes64 = s->d_nodes64; k.ref_anc = s->d_ref_anc; k.ref_depth = s->d_ref_depth; k.leaf_anc = s->d_leaf_anc;
  k.prim_ids = s->d_prim_ids; k.mats = s->d_mats; k.lights = s->d_lights;
  k.plane_boxes = s->d_plane_boxes; k.shade_recs = s->d_tri_normals; k.sky = s->d_sky; k.sky_u_tab = s->d_sky_u_tab; k.sky_u_q = s->d_sky_u_q ? s->d_sky_u_q + 2 : nullptr; k.sky_v_tab = s->d_sky_u_q ? s->d_sky_u_q + 2 + hs.sky_u_q.size() : nullptr;
  *out = s;
  return RT_OK;
}

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char *rt_build_id(void) { return RT_BUILD_ID; }
int rt_tile_skew(void) { return tile_skew(); }
const char *rt_last_error(void) { return g_err.c_str(); }

int rt_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device visible: the render core has no CPU fallback");
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) device = 0; }
  if (device >= n || device >= RT_MAX_DEVICES) return fail(RT_ERR_INVALID, "device %d out of range (%d visible)", device, n);
  HIP_TRY(hipSetDevice(device));
  int rc = init_device(device);
  if (rc) return rc;
  std::lock_guard<std::mutex> g(g_init_mu);
  g_group.assign(1, device);
  return RT_OK;
}

int rt_init_devices(int32_t n_devices, const int32_t *devices) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(RT_ERR_NO_DEVICE, "no HIP device visible: the render core has no CPU fallback");
  if (n_devices == 0 && !devices) n_devices = n;             // 0 = all visible devices
  if (n_devices <= 0 || n_devices > RT_MAX_DEVICES) return fail(RT_ERR_INVALID, "bad device count %d", (int)n_devices);
  std::vector<int> group(n_devices);
  for (int i = 0; i < n_devices; ++i) {
    group[i] = devices ? devices[i] : i;
    if (group[i] < 0 || group[i] >= n) return fail(RT_ERR_INVALID, "device %d out of range (%d visible)", group[i], n);
  }
  for (int i = n_devices - 1; i >= 0; --i) {                 // ends with the group's device 0 current
    HIP_TRY(hipSetDevice(group[i]));
    int rc = init_device(group[i]);
    if (rc) return rc;
    if (group[i] != group[0]) {                              // let the packed tiles go to device 0 directly over xGMI
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, group[i], group[0]) == hipSuccess && can) {
        hipError_t e = hipDeviceEnablePeerAccess(group[0], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // the copy then stages through the host
        else if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
      }
    }
  }
  std::lock_guard<std::mutex> g(g_init_mu);
  g_group = group;
  return RT_OK;
}

int rt_device_count(void) { return (int)group_snapshot().size(); }

void rt_shutdown(void) {
  std::lock_guard<std::mutex> g(g_init_mu);
  g_group.clear();
}

void rt_scene_destroy(rt_scene *s) {
  if (!s) return;
  int cur = 0;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;
  for (Replica *r : s->rep) replica_destroy(r);
  if (have_cur) (void)hipSetDevice(cur);
  delete s;
}

int rt_scene_create(const rt_scene_desc *d, rt_scene **out) {
  if (!d || !out) return fail(RT_ERR_INVALID, "null argument");
  *out = nullptr;
  int rc = ensure_init();
  if (rc) return rc;
  HostScene hs;
  if ((rc = prepare_host_scene(d, hs))) return rc;
  auto *s = new rt_scene();
  const std::vector<int> group = group_snapshot();           // (non-empty after ensure_init)
  if (group.empty()) return fail(RT_ERR_INVALID, "rt_shutdown ran concurrently with rt_scene_create");
  for (size_t i = group.size(); i-- > 0;) {                  // device 0 of the group last: it stays current
    Replica *r = nullptr;
    if (hipSetDevice(group[i]) != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_HIP, "hipSetDevice(%d) failed", group[i]); }
    if ((rc = upload_replica(hs, group[i], &r))) { rt_scene_destroy(s); return rc; }
    s->rep.insert(s->rep.begin(), r);
  }
  *out = s;
  return RT_OK;
}


int rt_fast_bvh_build(int32_t n_prims, const double *boxes, int32_t max_leaf, int32_t n_threads,
                      double *node_bounds, int32_t *node_child, int32_t *order, int32_t *depth) {
  if (n_prims <= 0 || !boxes || !node_bounds || !node_child || !order) return fail(RT_ERR_INVALID, "bad fast-tree arguments");
  for (size_t i = 0; i < (size_t)6 * n_prims; ++i)
    if (!std::isfinite(boxes[i])) return fail(RT_ERR_INVALID, "non-finite primitive box");
  FastBvh fb;
  build_fast_bvh(n_prims, boxes, std::max(1, max_leaf), 1.2, n_threads, fb);
  std::memcpy(node_bounds, fb.bounds.data(), fb.bounds.size() * sizeof(double));
  std::memcpy(node_child, fb.child.data(), fb.child.size() * sizeof(int32_t));
  std::memcpy(order, fb.order.data(), fb.order.size() * sizeof(int32_t));
  if (depth) *depth = fb.depth;
  return RT_OK;
}

int rt_tile_count(int32_t W, int32_t H, int32_t rank, int32_t world) {
  if (W <= 0 || H <= 0 || world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad tile query");
  int tx = tiles_x_of(W), ty = tiles_y_of(H), n = 0;
  for (int y = 0; y < ty; ++y) n += tiles_in_row(tx, y, rank, world, tile_skew());
  return n;
}

int rt_tile_list(int32_t W, int32_t H, int32_t rank, int32_t world, int32_t *tiles) {
  if (W <= 0 || H <= 0 || world <= 0 || rank < 0 || rank >= world || !tiles) return fail(RT_ERR_INVALID, "bad tile query");
  int tx = tiles_x_of(W), ty = tiles_y_of(H), n = 0;
  for (int y = 0; y < ty; ++y)
    for (int x = 0; x < tx; ++x)
      if (((x + (long long)tile_skew() * y) % world) == rank) tiles[n++] = y * tx + x;
  return n;
}

}  // extern "C"

static int check_frame(const rt_scene *s, const rt_camera *cam, const rt_params *p) {
  if (!s || s->rep.empty() || !cam || !p) return fail(RT_ERR_INVALID, "null argument");
  if (p->width <= 0 || p->height <= 0 || p->samples_per_pixel <= 0 || p->max_bounces < 0)
    return fail(RT_ERR_INVALID, "bad image/raytracer settings");
  if ((long long)p->width * p->height >= (1LL << 31)) return fail(RT_ERR_UNSUPPORTED, "image too large");
  if (p->rr_start_bounce < 0) return fail(RT_ERR_INVALID, "rr_start_bounce must be >= 0 (0 = off, the reference's behaviour)");
  if (p->box_precision < 0 || p->box_precision > 2) return fail(RT_ERR_INVALID, "box_precision must be 0, 1 or 2");
  return RT_OK;
}

// This rank's tile list, written on the device by k_make_tiles (no host list, no copy).  A growable workspace enlarges
// its buffer on demand; a fixed one refuses frames with more tiles than it holds.
static int ensure_tiles(Workspace &ws, int W, int H, int rank, int world, hipStream_t stream) {
  if (ws.tl_w == W && ws.tl_h == H && ws.tl_rank == rank && ws.tl_world == world && !ws.fixed) return RT_OK;
  const int n = rt_tile_count(W, H, rank, world);
  if (n < 0) return n;
  if ((size_t)n > ws.tiles_cap) {
    if (ws.fixed) return fail(RT_ERR_UNSUPPORTED, "%d tiles exceed the %zu the static scratch holds", n, ws.tiles_cap);
    HIP_TRY(hipStreamSynchronize(stream));
    (void)hipFree(ws.tiles); ws.tiles = nullptr; ws.tiles_cap = 0;
    if (hipMalloc((void **)&ws.tiles, (size_t)n * sizeof(int32_t)) != hipSuccess) return fail(RT_ERR_OOM, "tile list of %d entries", n);
    ws.tiles_cap = (size_t)n;
  }
  const int ty = tiles_y_of(H);
  if (n > 0) {
    hipLaunchKernelGGL(k_make_tiles, dim3((unsigned)((ty + 255) / 256)), dim3(256), 0, stream, tiles_x_of(W), ty, rank, world, tile_skew(), ws.tiles);
    HIP_TRY(hipGetLastError());
  }
  ws.tl_w = W; ws.tl_h = H; ws.tl_rank = rank; ws.tl_world = world; ws.tl_n = n;
  return RT_OK;
}

// One frame (or one rank's share of it) = one or more launches of k_render_tiles + k_resolve.  A launch covers a
// range of the rank's tiles whose per-path radiance fits the sample buffer; normally that is the whole frame
// (1080p x 100 spp: 5 GB of 288 GB).  The split never changes a result: k_resolve sums in sample order.
template <typename BT, bool STATS, bool FAST, bool RR = false>
static int launch_kernel(Replica *s, Workspace &ws, const KFrame &fr_in, hipStream_t stream) {
  KFrame fr = fr_in;
  const DevCtx &dc = g_dev[s->device];
  const int n_internal = FAST ? s->f_internal : s->n_internal;
  fr.stack_depth = std::max(1, FAST ? s->f_depth : s->depth) + 1;
  // One workgroup per CU, RT_MIN_WAVES waves per SIMD; deep trees (long per-lane stacks) get fewer waves so that the
  // stacks still fit the CU's LDS.  What is left of 160 KB caches the top BVH levels.
  // per wave: RT_STACK_LDS stack rows + the record's hot rows; deeper stack entries overflow to global memory
  const size_t stack_bytes = (size_t)(RT_STACK_LDS + LR_ROWS) * 64 * sizeof(uint2);
  fr.scratch_rows = PR_ROWS + std::max(0, fr.stack_depth - RT_STACK_LDS);
  fr.lit_row = -1;
  if (s->has_boxed_planes) {                                  // literal_hit's per-lane stack: 3 rows per level of the reference's tree
    fr.lit_row = fr.scratch_rows;
    fr.scratch_rows += 3 * (std::max(1, s->depth) + 1);
  }
  fr.exact_row = 0;
  if (sizeof(BT) == 8) { fr.exact_row = fr.scratch_rows; fr.scratch_rows += fr.stack_depth; }
  int wpb = RT_BLOCK_THREADS / 64;
  if (const char *env = getenv("RT_WAVES_PER_BLOCK")) wpb = std::max(4, std::min(wpb, atoi(env)));   // experiment knob
  const size_t lds_cu = 160 * 1024 / RT_BLOCKS_PER_CU;       // LDS one workgroup may use
#ifndef RT_LDS_RESERVE
#define RT_LDS_RESERVE (6 * 1024)
#endif
  while (wpb > 4 && (size_t)wpb * stack_bytes > lds_cu - RT_LDS_RESERVE) --wpb;
  const int threads = wpb * 64;
  const size_t lds_waves = (size_t)wpb * stack_bytes;
  fr.top_nodes = 0;
  if (sizeof(BT) == 4 && lds_waves < lds_cu - 4 * 1024) {
    size_t room = (lds_cu - 4 * 1024 - lds_waves) / sizeof(DNode<float>);
    int cap = RT_TOP_NODES_MAX;
    const char *env = getenv("RT_TOP_NODES");               // experiment knob
    if (env) cap = atoi(env);
    fr.top_nodes = (int)std::min<size_t>(std::min<size_t>(room, (size_t)cap), (size_t)n_internal);
  }
  size_t lds = lds_waves + (size_t)fr.top_nodes * sizeof(DNode<float>);
  if (const char *pad = getenv("RT_LDS_BYTES")) lds = std::max(lds, (size_t)atol(pad));   // experiment knob: caps blocks per CU
  if (lds > lds_cu) return fail(RT_ERR_UNSUPPORTED, "traversal stack needs %zu bytes of LDS", lds);
  auto kern = k_render_tiles<BT, STATS, FAST, RR>;
  if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, threads, lds));
  per_cu = std::max(1, per_cu);
  if (getenv("RT_DEBUG")) fprintf(stderr, "rtcore: %d threads/workgroup, %zu B LDS (%d cached nodes), %d workgroups/CU\n", threads, lds, fr.top_nodes, per_cu);

  // tiles per launch from the sample-buffer budget
  const size_t bytes_per_tile = (size_t)64 * fr.spp * 3 * sizeof(double);
  const size_t max_tiles_32 = (size_t)0x7fffffff / ((size_t)64 * fr.spp);     // item indices are 32-bit: larger frames take several launches
  if (max_tiles_32 == 0) return fail(RT_ERR_UNSUPPORTED, "samplesPerPixel %d too large", fr.spp);
  size_t tpp = std::min<size_t>((size_t)fr.n_tiles, max_tiles_32);
  if (tpp * bytes_per_tile > ws.samples_cap) {
    if (ws.fixed) {
      tpp = ws.samples_cap / bytes_per_tile;
      if (tpp == 0) return fail(RT_ERR_UNSUPPORTED, "samplesPerPixel %d: one tile's samples (%zu bytes) exceed the static scratch (%zu)", fr.spp, bytes_per_tile, ws.samples_cap);
    } else {
      size_t budget;
      if (const char *env = getenv("RT_SAMPLE_BUDGET_MB")) budget = (size_t)atol(env) << 20;   // tests: forces several launches
      else {
        size_t fre = 0, tot = 0;
        HIP_TRY(hipMemGetInfo(&fre, &tot));
        budget = std::min<size_t>((fre + ws.samples_cap) / 2, (size_t)64 << 30);
      }
      tpp = std::max<size_t>(1, std::min(tpp, budget / bytes_per_tile));
      if (tpp * bytes_per_tile > ws.samples_cap) {            // grows only on the first launch / a larger frame
        HIP_TRY(hipStreamSynchronize(stream));
        (void)hipFree(ws.samples); ws.samples = nullptr; ws.samples_cap = 0;
        if (hipMalloc((void **)&ws.samples, tpp * bytes_per_tile) != hipSuccess)
          return fail(RT_ERR_OOM, "sample buffer of %zu bytes (%zu tiles x %d spp)", tpp * bytes_per_tile, tpp, fr.spp);
        ws.samples_cap = tpp * bytes_per_tile;
      }
    }
  }
  fr.samples = ws.samples;

  const size_t waves_first = (tpp * 64 * (size_t)fr.spp + 63) / 64;           // wave-loads of items in the largest launch
  int grid_max = (int)std::max<size_t>(1, std::min<size_t>((waves_first + wpb - 1) / wpb, (size_t)dc.n_cu * per_cu));
  const size_t bytes_per_wave = (size_t)fr.scratch_rows * 64 * sizeof(double);
  if ((size_t)grid_max * wpb * bytes_per_wave > ws.fscratch_cap) {
    if (ws.fixed) {                                          // fewer resident workgroups: still correct, the static scratch just holds fewer records
      grid_max = (int)std::min<size_t>((size_t)grid_max, ws.fscratch_cap / ((size_t)wpb * bytes_per_wave));
      if (grid_max < 1) return fail(RT_ERR_UNSUPPORTED, "BVH depth %d: one workgroup's path records (%zu bytes) exceed the static scratch (%zu)",
                                    fr.stack_depth, (size_t)wpb * bytes_per_wave, ws.fscratch_cap);
    } else {                                                 // grows only on the first launch / a larger grid; never inside a steady-state loop
      HIP_TRY(hipStreamSynchronize(stream));
      (void)hipFree(ws.fscratch); ws.fscratch = nullptr; ws.fscratch_cap = 0;
      const size_t want = (size_t)grid_max * wpb * bytes_per_wave;
      if (hipMalloc((void **)&ws.fscratch, want) != hipSuccess) return fail(RT_ERR_OOM, "path records for %zu waves", (size_t)grid_max * wpb);
      ws.fscratch_cap = want;
    }
  }
  fr.fscratch = ws.fscratch;
  fr.tile_counter = ws.counter; fr.stats = ws.stats;
  if (STATS) HIP_TRY(hipMemsetAsync(ws.stats, 0, STAT_COUNT * sizeof(unsigned long long), stream));
  KScene ks = s->ks;
  if (sizeof(BT) == 8) ks.nodes = s->d_nodes64;
  if (FAST) { ks.nodes = s->d_fnodes32; ks.tprims = s->d_fprims; ks.root_ref = s->f_root_ref; }
  for (size_t t0 = 0; t0 < (size_t)fr.n_tiles; t0 += tpp) {
    fr.tile_begin = (int32_t)t0;
    fr.tile_end = (int32_t)std::min<size_t>((size_t)fr.n_tiles, t0 + tpp);
    const size_t slots = (size_t)(fr.tile_end - fr.tile_begin) * 64;
    fr.n_items = (uint32_t)(slots * fr.spp);
    // chunk: at most 1/25 of what one resident wave will process, a power of two in [64, 512] (see refill in the kernel)
    {
      const size_t per_wave = (size_t)fr.n_items / ((size_t)grid_max * wpb);
      uint32_t ch = 64;
      while (ch < 512 && (size_t)ch * 2 * RT_CHUNKS_PER_WAVE <= per_wave) ch *= 2;
      if (const char *env = getenv("RT_CHUNK_ITEMS")) ch = (uint32_t)std::max(64, (atoi(env) / 64) * 64);   // experiment knob
      fr.chunk = ch;
    }
    const size_t want = ((size_t)fr.n_items / 64 + wpb - 1) / wpb;
    const int grid = (int)std::max<size_t>(1, std::min<size_t>(want, (size_t)grid_max));
    HIP_TRY(hipMemsetAsync(ws.counter, 0, sizeof(uint32_t), stream));
    const bool timed = !ws.fixed && s->kt_pending < Replica::KT_RING;      // (creating an event allocates: not on the fixed workspace)
    if (timed) {
      while ((int)s->kt_events.size() < 2 * (s->kt_pending + 1)) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->kt_events.push_back(e);
      }
      HIP_TRY(hipEventRecord(s->kt_events[2 * s->kt_pending], stream));
    }
#ifdef RT_TIMERS
    static unsigned long long *d_dbg = nullptr;            // experiment build: per-wave timeline, dumped to $RT_WAVE_DUMP after each launch
    const char *dump = getenv("RT_WAVE_DUMP");
    if (dump && !d_dbg) HIP_TRY(hipMalloc((void **)&d_dbg, ((size_t)dc.n_cu * 64 * 4 + 2 * 262144) * sizeof(unsigned long long)));
    fr.dbg = dump ? d_dbg : nullptr;
    if (dump) HIP_TRY(hipMemsetAsync(d_dbg, 0, ((size_t)grid * wpb * 4 + 2 * 262144) * sizeof(unsigned long long), stream));
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, ks, fr);
    HIP_TRY(hipGetLastError());
#ifdef RT_TIMERS
    if (dump) {
      std::vector<unsigned long long> h((size_t)grid * wpb * 4 + 2 * 262144);
      HIP_TRY(hipMemcpy(h.data(), d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
      if (FILE *f = fopen(dump, "wb")) { fwrite(h.data(), sizeof(unsigned long long), h.size(), f); fclose(f); }
    }
#endif
    if (timed) { HIP_TRY(hipEventRecord(s->kt_events[2 * s->kt_pending + 1], stream)); ++s->kt_pending; }
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((slots / 64 + RS_WAVES - 1) / RS_WAVES)), dim3(64 * RS_WAVES), 0, stream, fr);
    HIP_TRY(hipGetLastError());
  }
  return RT_OK;
}

static int launch_render(Replica *s, Workspace &ws, const rt_camera *cam, const rt_params *p, int rank, int world, bool packed,
                         double *d_color, double *d_radiance, int32_t *d_primid, uint8_t *d_rgb8, hipStream_t stream) {
  int rc;
  KFrame fr{};
  fr.cam = *cam;
  fr.W = p->width; fr.H = p->height; fr.spp = p->samples_per_pixel; fr.aa = p->antialiasing != 0; fr.maxb = p->max_bounces;
  fr.exposure = p->exposure; fr.gamma = p->gamma;
  fr.rr_start = p->rr_start_bounce;
  fr.key0 = (uint32_t)p->seed; fr.key1 = (uint32_t)(p->seed >> 32);
  fr.tiles_x = tiles_x_of(p->width);
  if (world > 1) {
    if ((rc = ensure_tiles(ws, p->width, p->height, rank, world, stream))) return rc;
    fr.tiles = ws.tiles; fr.n_tiles = ws.tl_n;
  } else {                                                   // one rank: its tile list is 0, 1, 2, ... - no list needed
    fr.tiles = nullptr; fr.n_tiles = fr.tiles_x * tiles_y_of(p->height);
  }
  fr.packed = packed ? 1 : 0;
  fr.out_color = d_color; fr.out_radiance = d_radiance; fr.out_primid = d_primid; fr.out_rgb8 = d_rgb8;
  if (fr.n_tiles == 0) return RT_OK;
#ifdef RT_TIMERS
  const bool st = false;                                    // experiment build: collect_stats times the bodies of the TIMED variant
  if (p->collect_stats) HIP_TRY(hipMemsetAsync(ws.stats, 0, STAT_COUNT * sizeof(unsigned long long), stream));
#else
  const bool st = p->collect_stats != 0;
#endif
  const int mode = s->inverted_boxes ? 1 : p->box_precision;
  if (fr.rr_start > 0) {                                    // the opt-in extension has its own instantiations
    if (mode == 1) return st ? launch_kernel<double, true, false, true>(s, ws, fr, stream) : launch_kernel<double, false, false, true>(s, ws, fr, stream);
    if (mode == 0 && s->has_fast) return st ? launch_kernel<float, true, true, true>(s, ws, fr, stream) : launch_kernel<float, false, true, true>(s, ws, fr, stream);
    return st ? launch_kernel<float, true, false, true>(s, ws, fr, stream) : launch_kernel<float, false, false, true>(s, ws, fr, stream);
  }
  if (mode == 1) return st ? launch_kernel<double, true, false>(s, ws, fr, stream) : launch_kernel<double, false, false>(s, ws, fr, stream);
  if (mode == 0 && s->has_fast) return st ? launch_kernel<float, true, true>(s, ws, fr, stream) : launch_kernel<float, false, true>(s, ws, fr, stream);
  return st ? launch_kernel<float, true, false>(s, ws, fr, stream) : launch_kernel<float, false, false>(s, ws, fr, stream);
}

// Template arguments of the k_render_tiles instantiation launch_render picks for (scene, params), as rocprofv3 prints them.
static std::string kernel_args_of(const Replica *s, const rt_params *p) {
  const int mode = s->inverted_boxes ? 1 : p->box_precision;
  const bool fast = mode == 0 && s->has_fast;
  char buf[64];
  snprintf(buf, sizeof buf, "%s, %s, %s, %s", mode == 1 ? "double" : "float", p->collect_stats ? "true" : "false",
           fast ? "true" : "false", p->rr_start_bounce > 0 ? "true" : "false");
  return buf;
}

static int read_stats(const Workspace &ws, rt_stats *st) {   // ADDS this workspace's counters to *st
  unsigned long long h[STAT_COUNT];
  HIP_TRY(hipMemcpy(h, ws.stats, sizeof h, hipMemcpyDeviceToHost));
  st->rays_closest += h[STAT_CLOSEST]; st->rays_shadow += h[STAT_SHADOW]; st->nodes_visited += h[STAT_NODES];
  st->tri_tests += h[STAT_TRI]; st->sphere_tests += h[STAT_SPHERE]; st->plane_tests += h[STAT_PLANE]; st->shaded_hits += h[STAT_SHADED];
  st->sched_node_steps += h[STAT_IT_NODE] / 64; st->sched_prim_steps += h[STAT_IT_PRIM] / 64; st->sched_shade_steps += h[STAT_IT_SHADE] / 64;
  st->sched_live_lanes += h[STAT_LIVE];
  st->top_node_hits += h[STAT_TOPHIT];
  st->rays_shadow_culled += h[STAT_SHADOW_CULLED];
  st->rays_literal += h[STAT_LITERAL];
  for (int k = 0; k < 5; ++k) { st->shade_lanes[k] += h[STAT_SH_L0 + k]; st->shade_trips[k] += h[STAT_SH_W0 + k]; }
  for (int k = 0; k < 7; ++k) st->cyc_shade_detail[k] += h[STAT_T_SH0 + k];
  st->cyc_vote += h[STAT_T_VOTE]; st->cyc_node += h[STAT_T_NODE]; st->cyc_prim += h[STAT_T_PRIM]; st->cyc_shade += h[STAT_T_SHADE]; st->cyc_refill += h[STAT_T_REFILL];
  return RT_OK;
}

// device-resident tile lists for rt_untile_device, created once per (device, W, H, rank, world) and kept
struct TileCacheEntry { int device, W, H, rank, world, n; int32_t *d_tiles; };
static std::vector<TileCacheEntry> g_tile_cache;
static std::mutex g_tile_mu;
static int cached_tiles(int W, int H, int rank, int world, const int32_t **d_tiles, int *n, hipStream_t stream) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(g_tile_mu);
  for (auto &e : g_tile_cache)
    if (e.device == dev && e.W == W && e.H == H && e.rank == rank && e.world == world) { *d_tiles = e.d_tiles; *n = e.n; return RT_OK; }
  int cnt = rt_tile_count(W, H, rank, world);
  if (cnt < 0) return cnt;
  int32_t *d = nullptr;
  if (hipMalloc((void **)&d, (size_t)std::max(1, cnt) * sizeof(int32_t)) != hipSuccess) return fail(RT_ERR_OOM, "tile list of %d entries", cnt);
  if (cnt > 0) {
    const int ty = tiles_y_of(H);
    hipLaunchKernelGGL(k_make_tiles, dim3((unsigned)((ty + 255) / 256)), dim3(256), 0, stream, tiles_x_of(W), ty, rank, world, tile_skew(), d);
    HIP_TRY(hipGetLastError());
    if (stream != nullptr) HIP_TRY(hipStreamSynchronize(stream));   // later callers may use another stream
    else HIP_TRY(hipDeviceSynchronize());
  }
  g_tile_cache.push_back({dev, W, H, rank, world, cnt, d});
  *d_tiles = d; *n = cnt;
  return RT_OK;
}

static int untile_launch(int W, int H, int rank, int world, const void *d_packed, void *d_image, int n_comp, bool bytes, hipStream_t stream) {
  int n = 0, rc;
  const int32_t *d_t = nullptr;
  if ((rc = cached_tiles(W, H, rank, world, &d_t, &n, stream))) return rc;
  if (n == 0) return RT_OK;
  const size_t slots = (size_t)n * 64;
  if (bytes)
    hipLaunchKernelGGL(k_untile_bytes, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream,
                       d_t, n, tiles_x_of(W), W, H, n_comp, (const uint8_t *)d_packed, (uint8_t *)d_image);
  else
    hipLaunchKernelGGL(k_untile, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, stream,
                       d_t, n, tiles_x_of(W), W, H, n_comp, (const double *)d_packed, (double *)d_image);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

// per (device, W, H, world): k_make_row_base's table for k_untile_all, created once and kept
struct RowBaseEntry { int device, W, H, world; int32_t *d_base; };
static std::vector<RowBaseEntry> g_rowbase_cache;
static int untile_all_launch(int W, int H, int world, const void *d_packed_all, size_t rank_stride_bytes, void *d_image, int n_bytes, hipStream_t stream) {
  if (W <= 0 || H <= 0 || world <= 0) return fail(RT_ERR_INVALID, "bad untile arguments");
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  const int tx = tiles_x_of(W), ty = tiles_y_of(H);
  int32_t *d_base = nullptr;
  {
    std::lock_guard<std::mutex> g(g_tile_mu);
    for (auto &e : g_rowbase_cache) if (e.device == dev && e.W == W && e.H == H && e.world == world) d_base = e.d_base;
    if (!d_base) {
      if (hipMalloc((void **)&d_base, (size_t)world * ty * sizeof(int32_t)) != hipSuccess) return fail(RT_ERR_OOM, "row-base table of %d entries", world * ty);
      hipLaunchKernelGGL(k_make_row_base, dim3((unsigned)((world * ty + 255) / 256)), dim3(256), 0, stream, tx, ty, world, tile_skew(), d_base);
      HIP_TRY(hipGetLastError());
      if (stream != nullptr) HIP_TRY(hipStreamSynchronize(stream));   // later callers may use another stream
      else HIP_TRY(hipDeviceSynchronize());
      g_rowbase_cache.push_back({dev, W, H, world, d_base});
    }
  }
  const size_t npx = (size_t)W * H;
  hipLaunchKernelGGL(k_untile_all, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, stream, d_base, tx, ty, world, tile_skew(), W, H, n_bytes,
                     rank_stride_bytes, (const uint8_t *)d_packed_all, (uint8_t *)d_image);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

static int ensure_frame_buffers(Replica *s, size_t npx) {
  if (npx <= s->cap_px) return RT_OK;
  (void)hipFree(s->d_color); (void)hipFree(s->d_radiance); (void)hipFree(s->d_primid); (void)hipFree(s->d_rgb8);
  s->d_color = s->d_radiance = nullptr; s->d_primid = nullptr; s->d_rgb8 = nullptr; s->cap_px = 0;
  if (hipMalloc((void **)&s->d_color, npx * 24) != hipSuccess || hipMalloc((void **)&s->d_radiance, npx * 24) != hipSuccess ||
      hipMalloc((void **)&s->d_primid, npx * 4) != hipSuccess || hipMalloc((void **)&s->d_rgb8, npx * 3) != hipSuccess)
    return fail(RT_ERR_OOM, "output buffers for %zu pixels", npx);
  s->cap_px = npx;
  return RT_OK;
}

// One frame on one device: the whole image, row-major, then to the host.
static int render_single(Replica *s, const rt_camera *cam, const rt_params *p, rt_output *out, rt_stats *stats) {
  int rc;
  HIP_TRY(hipSetDevice(s->device));
  const size_t npx = (size_t)p->width * p->height;
  if ((rc = ensure_frame_buffers(s, npx))) return rc;
  hipStream_t stream = nullptr;
  HIP_TRY(hipEventRecord(s->ev0, stream));
  // (k_resolve writes every requested form of the frame in its one pass, the 8-bit one included)
  rc = launch_render(s, s->ws, cam, p, 0, 1, false, out->color ? s->d_color : nullptr, out->radiance ? s->d_radiance : nullptr,
                     out->prim_id ? s->d_primid : nullptr, out->rgb8 ? s->d_rgb8 : nullptr, stream);
  if (rc) return rc;
  HIP_TRY(hipEventRecord(s->ev1, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (out->color) HIP_TRY(hipMemcpy(out->color, s->d_color, npx * 24, hipMemcpyDeviceToHost));
  if (out->radiance) HIP_TRY(hipMemcpy(out->radiance, s->d_radiance, npx * 24, hipMemcpyDeviceToHost));
  if (out->prim_id) HIP_TRY(hipMemcpy(out->prim_id, s->d_primid, npx * 4, hipMemcpyDeviceToHost));
  if (out->rgb8) HIP_TRY(hipMemcpy(out->rgb8, s->d_rgb8, npx * 3, hipMemcpyDeviceToHost));
  if (stats) {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
    stats->ms_render = ms;
    if (p->collect_stats && (rc = read_stats(s->ws, stats))) return rc;
  }
  return RT_OK;
}

// One frame on a device group (rt_init_devices): replica i renders the tiles of rank i of world N from its own copy of
// the scene, on its own stream, driven by its own host thread; its packed tiles go to the group's device 0 as peer
// copies (one xGMI link per peer, all at once: a gather, not a ring) into ONE contiguous gather area per output, are un-tiled
// there by one launch per output (k_untile_all) and leave for the host in one copy per output.  Every replica's k_resolve writes
// the 8-bit form of its tiles itself, so the 8-bit frame crosses xGMI as 3 bytes per pixel and no quantise launch exists.
// The frame equals the single-device frame bit for bit (RNG keyed on the global pixel).
static int render_group(rt_scene *sc, const rt_camera *cam, const rt_params *p, rt_output *out, rt_stats *stats) {
  const int N = (int)sc->rep.size();
  const int W = p->width, H = p->height;
  const size_t npx = (size_t)W * H;
  Replica *r0 = sc->rep[0];
  const bool want_color = out->color != nullptr, want_rad = out->radiance != nullptr, want_id = out->prim_id != nullptr;
  int rc;
  // ---- buffers: packed tiles on each replica's device, their landing place on device 0 (slice i of replica 0's contiguous
  //      gather areas), the frame on device 0
  size_t max_slots = 0;
  std::vector<int> n_tiles(N);
  for (int i = 0; i < N; ++i) { n_tiles[i] = rt_tile_count(W, H, i, N); max_slots = std::max(max_slots, (size_t)n_tiles[i] * 64); }
  HIP_TRY(hipSetDevice(r0->device));
  if ((rc = ensure_frame_buffers(r0, npx))) return rc;
  if (max_slots > r0->cap_slots || r0->G_n != N) {
    const size_t m = std::max<size_t>(max_slots, 64);
    for (int i = 1; i < N; ++i) {                            // the other replicas' own packed buffers
      Replica *r = sc->rep[i];
      HIP_TRY(hipSetDevice(r->device));
      (void)hipFree(r->p_color); (void)hipFree(r->p_radiance); (void)hipFree(r->p_primid); (void)hipFree(r->p_rgb8);
      r->p_color = r->p_radiance = nullptr; r->p_primid = nullptr; r->p_rgb8 = nullptr; r->cap_slots = 0;
      if (hipMalloc((void **)&r->p_color, m * 24) != hipSuccess || hipMalloc((void **)&r->p_radiance, m * 24) != hipSuccess ||
          hipMalloc((void **)&r->p_primid, m * 4) != hipSuccess || hipMalloc((void **)&r->p_rgb8, m * 3) != hipSuccess)
        return fail(RT_ERR_OOM, "packed tile buffers for %zu slots on device %d", m, r->device);
      r->cap_slots = m;
    }
    HIP_TRY(hipSetDevice(r0->device));
    (void)hipFree(r0->G_color); (void)hipFree(r0->G_radiance); (void)hipFree(r0->G_primid); (void)hipFree(r0->G_rgb8);
    r0->G_color = r0->G_radiance = nullptr; r0->G_primid = nullptr; r0->G_rgb8 = nullptr; r0->cap_slots = 0; r0->G_n = 0;
    if (hipMalloc((void **)&r0->G_color, (size_t)N * m * 24) != hipSuccess || hipMalloc((void **)&r0->G_radiance, (size_t)N * m * 24) != hipSuccess ||
        hipMalloc((void **)&r0->G_primid, (size_t)N * m * 4) != hipSuccess || hipMalloc((void **)&r0->G_rgb8, (size_t)N * m * 3) != hipSuccess)
      return fail(RT_ERR_OOM, "gather areas for %d x %zu slots on device %d", N, m, r0->device);
    r0->cap_slots = m; r0->G_n = N;
    for (int i = 0; i < N; ++i) {
      Replica *r = sc->rep[i];
      r->g_color = r0->G_color + (size_t)i * m * 3; r->g_radiance = r0->G_radiance + (size_t)i * m * 3;
      r->g_primid = r0->G_primid + (size_t)i * m; r->g_rgb8 = r0->G_rgb8 + (size_t)i * m * 3;
    }
    r0->p_color = r0->g_color; r0->p_radiance = r0->g_radiance; r0->p_primid = r0->g_primid; r0->p_rgb8 = r0->g_rgb8;
  }
  const size_t m_slots = r0->cap_slots;
  for (int i = 0; i < N; ++i) {
    Replica *r = sc->rep[i];
    if (!r->stream) { HIP_TRY(hipSetDevice(r->device)); HIP_TRY(hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking)); }
  }
  // ---- fan out: one host thread per device (replaces the 24 green threads + TQueue of ImageGenerator.hs:37,64-94)
  std::vector<int> rcs(N, RT_OK);
  std::vector<std::string> errs(N);
  auto work = [&](int i) {
    Replica *r = sc->rep[i];
    auto body = [&]() -> int {
      HIP_TRY(hipSetDevice(r->device));
      if (n_tiles[i] == 0) return RT_OK;
      const size_t slots = (size_t)n_tiles[i] * 64;
      HIP_TRY(hipEventRecord(r->ev0, r->stream));
      // k_resolve writes every requested form of the packed tiles in its one pass; the 8-bit frame crosses xGMI as 3 bytes per pixel
      int rc2 = launch_render(r, r->ws, cam, p, i, N, true, want_color ? r->p_color : nullptr, want_rad ? r->p_radiance : nullptr,
                              want_id ? r->p_primid : nullptr, out->rgb8 ? r->p_rgb8 : nullptr, r->stream);
      if (rc2) return rc2;
      HIP_TRY(hipEventRecord(r->ev1, r->stream));
      if (i > 0) {                                           // the exchange step: packed tiles to device 0
        if (out->rgb8) HIP_TRY(hipMemcpyPeerAsync(r->g_rgb8, r0->device, r->p_rgb8, r->device, slots * 3, r->stream));
        if (want_color) HIP_TRY(hipMemcpyPeerAsync(r->g_color, r0->device, r->p_color, r->device, slots * 24, r->stream));
        if (want_rad) HIP_TRY(hipMemcpyPeerAsync(r->g_radiance, r0->device, r->p_radiance, r->device, slots * 24, r->stream));
        if (want_id) HIP_TRY(hipMemcpyPeerAsync(r->g_primid, r0->device, r->p_primid, r->device, slots * 4, r->stream));
      }
      HIP_TRY(hipStreamSynchronize(r->stream));
      return RT_OK;
    };
    rcs[i] = body();
    if (rcs[i]) errs[i] = g_err;                             // g_err is thread-local: hand the text to the calling thread
  };
  {
    std::vector<std::thread> th;
    for (int i = 1; i < N; ++i) th.emplace_back(work, i);
    work(0);
    for (auto &t : th) t.join();
  }
  for (int i = 0; i < N; ++i) if (rcs[i]) { g_err = errs[i]; return rcs[i]; }
  // ---- assemble on device 0: one launch per requested output scatters the tiles of all replicas
  HIP_TRY(hipSetDevice(r0->device));
  hipStream_t st0 = r0->stream;
  if (out->rgb8 && (rc = untile_all_launch(W, H, N, r0->G_rgb8, m_slots * 3, r0->d_rgb8, 3, st0))) return rc;
  if (want_color && (rc = untile_all_launch(W, H, N, r0->G_color, m_slots * 24, r0->d_color, 24, st0))) return rc;
  if (want_rad && (rc = untile_all_launch(W, H, N, r0->G_radiance, m_slots * 24, r0->d_radiance, 24, st0))) return rc;
  if (want_id && (rc = untile_all_launch(W, H, N, r0->G_primid, m_slots * 4, r0->d_primid, 4, st0))) return rc;
  HIP_TRY(hipStreamSynchronize(st0));
  if (out->color) HIP_TRY(hipMemcpy(out->color, r0->d_color, npx * 24, hipMemcpyDeviceToHost));
  if (out->radiance) HIP_TRY(hipMemcpy(out->radiance, r0->d_radiance, npx * 24, hipMemcpyDeviceToHost));
  if (out->prim_id) HIP_TRY(hipMemcpy(out->prim_id, r0->d_primid, npx * 4, hipMemcpyDeviceToHost));
  if (out->rgb8) HIP_TRY(hipMemcpy(out->rgb8, r0->d_rgb8, npx * 3, hipMemcpyDeviceToHost));
  if (stats) {
    for (int i = 0; i < N; ++i) {
      Replica *r = sc->rep[i];
      if (n_tiles[i] == 0) continue;
      HIP_TRY(hipSetDevice(r->device));
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, r->ev0, r->ev1));
      stats->ms_render = std::max(stats->ms_render, (double)ms);   // the slowest device sets the frame
      if (p->collect_stats && (rc = read_stats(r->ws, stats))) return rc;
    }
    HIP_TRY(hipSetDevice(r0->device));
  }
  return RT_OK;
}

extern "C" {

int rt_render(rt_scene *s, const rt_camera *cam, const rt_params *p, rt_output *out, rt_stats *stats) {
  int rc = check_frame(s, cam, p);
  if (rc) return rc;
  if (!out) return fail(RT_ERR_INVALID, "null output");
  auto t0 = std::chrono::steady_clock::now();
  if (stats) { std::memset(stats, 0, sizeof *stats); stats->paths = (uint64_t)p->width * p->height * p->samples_per_pixel; }
  rc = s->rep.size() == 1 ? render_single(s->rep[0], cam, p, out, stats) : render_group(s, cam, p, out, stats);
  if (rc) return rc;
  if (stats) stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return RT_OK;
}

static int render_tiles_common(rt_scene *sc, Workspace &ws, const rt_camera *cam, const rt_params *p, int32_t rank, int32_t world,
                               void *d_color, void *d_radiance, void *d_prim_id, void *d_rgb8, void *stream, rt_stats *stats) {
  Replica *s = sc->rep[0];
  int rc = launch_render(s, ws, cam, p, rank, world, true, (double *)d_color, (double *)d_radiance, (int32_t *)d_prim_id, (uint8_t *)d_rgb8, (hipStream_t)stream);
  if (rc) return rc;
  if (stats) {
    std::memset(stats, 0, sizeof *stats);
    stats->paths = (uint64_t)rt_tile_count(p->width, p->height, rank, world) * 64 * p->samples_per_pixel;
    if (p->collect_stats) {
      HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
      if ((rc = read_stats(ws, stats))) return rc;
    }
  }
  return RT_OK;
}

int rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_params *p, int32_t rank, int32_t world,
                           void *d_color, void *d_radiance, void *d_prim_id, void *stream, rt_stats *stats) {
  int rc = check_frame(s, cam, p);
  if (rc) return rc;
  if (world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad rank/world");
  if (!d_color && !d_radiance && !d_prim_id) return fail(RT_ERR_INVALID, "no output buffer");
  HIP_TRY(hipSetDevice(s->rep[0]->device));
  return render_tiles_common(s, s->rep[0]->ws, cam, p, rank, world, d_color, d_radiance, d_prim_id, nullptr, stream, stats);
}

int rt_render_tiles_rgb8_device(rt_scene *s, const rt_camera *cam, const rt_params *p, int32_t rank, int32_t world,
                                void *d_rgb8, void *stream, rt_stats *stats) {
  int rc = check_frame(s, cam, p);
  if (rc) return rc;
  if (world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad rank/world");
  if (!d_rgb8) return fail(RT_ERR_INVALID, "no output buffer");
  HIP_TRY(hipSetDevice(s->rep[0]->device));
  return render_tiles_common(s, s->rep[0]->ws, cam, p, rank, world, nullptr, nullptr, nullptr, d_rgb8, stream, stats);
}

// The same frame share with NO allocation anywhere on the way: the launch workspace is the static device array
// `scratch` (through the per-device address resolved in init_device), the tile list is written by a kernel, no HIP
// event is created.  A frame larger than the sample area takes several launches; what does not fit at all is refused.
int rt_render_tiles_device_noalloc(rt_scene *s, const rt_camera *cam, const rt_params *p, int32_t rank, int32_t world,
                                   void *d_color, void *d_radiance, void *d_prim_id, void *stream, rt_stats *stats) {
  int rc = check_frame(s, cam, p);
  if (rc) return rc;
  if (world <= 0 || rank < 0 || rank >= world) return fail(RT_ERR_INVALID, "bad rank/world");
  if (!d_color && !d_radiance && !d_prim_id) return fail(RT_ERR_INVALID, "no output buffer");
  Replica *r = s->rep[0];
  HIP_TRY(hipSetDevice(r->device));
  unsigned char *base = g_dev[r->device].scratch;            // the device address of `scratch` on THIS GPU (never the symbol itself)
  if (!g_dev[r->device].ready || !base) return fail(RT_ERR_HIP, "static scratch of device %d was not resolved by rt_init", r->device);
  Workspace ws;
  ws.fixed = true;
  ws.counter = (uint32_t *)base;
  ws.stats = (unsigned long long *)(base + 256);
  ws.tiles = (int32_t *)(base + RT_SCRATCH_TILES_OFF); ws.tiles_cap = RT_SCRATCH_MAX_TILES;
  ws.fscratch = (double *)(base + RT_SCRATCH_REC_OFF); ws.fscratch_cap = RT_SCRATCH_SAMPLES_OFF - RT_SCRATCH_REC_OFF;
  ws.samples = (double *)(base + RT_SCRATCH_SAMPLES_OFF); ws.samples_cap = RT_SCRATCH_BYTES - RT_SCRATCH_SAMPLES_OFF;
  static_assert(256 + STAT_COUNT * sizeof(unsigned long long) <= RT_SCRATCH_TILES_OFF, "counters overlap the tile list");
  static_assert(RT_SCRATCH_REC_OFF % 256 == 0 && RT_SCRATCH_REC_OFF < RT_SCRATCH_SAMPLES_OFF, "scratch layout");
  // every call shares this GPU's array and finds the previous call's contents in it: the counter is cleared per launch
  // (launch_kernel), the statistics when they are collected, the tile list is rewritten, records and samples are
  // written before they are read
  return render_tiles_common(s, ws, cam, p, rank, world, d_color, d_radiance, d_prim_id, nullptr, stream, stats);
}

int rt_kernel_name(rt_scene *sc, const rt_params *p, char *buf, int32_t cap) {
  if (!sc || sc->rep.empty() || !p || !buf || cap <= 0) return fail(RT_ERR_INVALID, "null argument");
  const std::string a = "k_render_tiles<" + kernel_args_of(sc->rep[0], p) + ">";
  if ((size_t)cap <= a.size()) return fail(RT_ERR_INVALID, "buffer of %d bytes too small for the kernel name", (int)cap);
  std::memcpy(buf, a.c_str(), a.size() + 1);
  return RT_OK;
}

int rt_kernel_times(rt_scene *sc, double *sum_ms, int32_t *n_launches) {
  if (!sc || sc->rep.empty() || !sum_ms || !n_launches) return fail(RT_ERR_INVALID, "null argument");
  double worst = 0;
  int n0 = 0;
  for (size_t k = 0; k < sc->rep.size(); ++k) {
    Replica *s = sc->rep[k];
    HIP_TRY(hipSetDevice(s->device));
    double sum = 0;
    for (int i = 0; i < s->kt_pending; ++i) {
      float ms = 0;
      HIP_TRY(hipEventSynchronize(s->kt_events[2 * i + 1]));
      HIP_TRY(hipEventElapsedTime(&ms, s->kt_events[2 * i], s->kt_events[2 * i + 1]));
      sum += ms;
    }
    if (k == 0) n0 = s->kt_pending;
    worst = std::max(worst, sum);
    s->kt_pending = 0;
  }
  HIP_TRY(hipSetDevice(sc->rep[0]->device));
  *sum_ms = worst; *n_launches = n0;
  return RT_OK;
}

int rt_untile_device(int32_t W, int32_t H, int32_t rank, int32_t world, const void *d_packed, void *d_image,
                     int32_t n_comp, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!d_packed || !d_image || n_comp <= 0) return fail(RT_ERR_INVALID, "bad untile arguments");
  return untile_launch(W, H, rank, world, d_packed, d_image, n_comp, false, (hipStream_t)stream);
}

int rt_untile_bytes_device(int32_t W, int32_t H, int32_t rank, int32_t world, const void *d_packed, void *d_image,
                           int32_t n_bytes, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!d_packed || !d_image || n_bytes <= 0) return fail(RT_ERR_INVALID, "bad untile arguments");
  return untile_launch(W, H, rank, world, d_packed, d_image, n_bytes, true, (hipStream_t)stream);
}

int rt_untile_all_bytes_device(int32_t W, int32_t H, int32_t world, const void *d_packed_all, int64_t rank_stride_bytes,
                               void *d_image, int32_t n_bytes, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!d_packed_all || !d_image || n_bytes <= 0 || rank_stride_bytes < 0) return fail(RT_ERR_INVALID, "bad untile arguments");
  for (int r = 0; r < world; ++r)
    if ((int64_t)rt_tile_count(W, H, r, world) * 64 * n_bytes > rank_stride_bytes)
      return fail(RT_ERR_INVALID, "rank %d's %d tiles do not fit a stride of %lld bytes", r, rt_tile_count(W, H, r, world), (long long)rank_stride_bytes);
  return untile_all_launch(W, H, world, d_packed_all, (size_t)rank_stride_bytes, d_image, n_bytes, (hipStream_t)stream);
}

int rt_quantize_device(const void *d_color, void *d_rgb8, int64_t n_values, void *stream) {
  int rc = ensure_init();
  if (rc) return rc;
  if (!d_color || !d_rgb8 || n_values < 0) return fail(RT_ERR_INVALID, "bad quantize arguments");
  if (n_values == 0) return RT_OK;
  hipLaunchKernelGGL(k_quantize, dim3((unsigned)((n_values + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const double *)d_color, (uint8_t *)d_rgb8, (long long)n_values);
  HIP_TRY(hipGetLastError());
  return RT_OK;
}

}  // extern "C"
