// rthost_bvh.cpp — host-side mirror of the reference's BVH construction and camera set-up
// (include/rthost.h).  Index-based and thread-parallel, but leaf-for-leaf identical to
//   constructBVHWithLimit   /root/reference/app/Hittable/BVH.hs:29-48
// i.e. same leaf rule (objCount <= leafThreshold || depth >= maxDepth, :36), same axis choice
// (:51-71), same STABLE sort on the AABB centroid (:74-80), same split at n `div` 2 (:42), boxes
// as exact min/max unions (:101-103).  Output is the pre-order flattening rt_scene_desc expects.
//
// Compile with -ffp-contract=off: the centroid key 0.5*(lo+hi) must round like the reference's.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <future>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/rthost.h"

namespace rth {
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
}  // namespace rth
using rth::fail;

namespace {

struct Box { double lo[3], hi[3]; };

// boundingBox of each primitive kind (Triangle.hs:24-28, Sphere.hs:20-22, Plane.hs:20)
Box prim_box(int kind, const double *d) {
  Box b;
  if (kind == RT_PRIM_TRIANGLE) {
    for (int k = 0; k < 3; ++k) {
      double mn = std::min(std::min(d[k], d[3 + k]), d[6 + k]);
      double mx = std::max(std::max(d[k], d[3 + k]), d[6 + k]);
      b.lo[k] = mn - 1e-8; b.hi[k] = mx + 1e-8;      // pad = epsilon, Utils/Constants.hs:15-16
    }
  } else if (kind == RT_PRIM_SPHERE) {
    for (int k = 0; k < 3; ++k) { b.lo[k] = d[k] - d[3]; b.hi[k] = d[k] + d[3]; }
  } else {
    b.lo[0] = -1e6; b.lo[1] = -1e-3; b.lo[2] = -1e6; b.hi[0] = 1e6; b.hi[1] = 1e-3; b.hi[2] = 1e6;
  }
  return b;
}

// number of nodes of the subtree the reference builds for `count` objects at `depth`
long subtree_nodes(long count, int depth, int leaf_thr, int max_depth) {
  if (count <= leaf_thr || depth >= max_depth) return 1;
  long l = count / 2;
  return 1 + subtree_nodes(l, depth + 1, leaf_thr, max_depth) + subtree_nodes(count - l, depth + 1, leaf_thr, max_depth);
}

struct Task { long b, e; int depth; long node, prim0; };

}  // namespace

struct rth_bvh {
  std::vector<double> node_bounds;
  std::vector<int32_t> node_child;
  std::vector<int32_t> prim_kind, prim_material, prim_id;
  std::vector<double> prim_data;
  int n_nodes = 0, n_leaves = 0, depth = 0;
};

extern "C" {

const char *rth_last_error(void) { return rth::g_err.c_str(); }

int rth_bvh_build(int32_t n, const int32_t *kind, const double *data, const int32_t *material,
                  int32_t leaf_thr, int32_t max_depth, int32_t n_threads, rth_bvh **out) {
  if (!out) return fail(RT_ERR_INVALID, "null output");
  *out = nullptr;
  if (n <= 0) return fail(RT_ERR_EMPTY_SCENE, "Cannot compute bounding box of empty list");   // BVH.hs:102
  if (!kind || !data || !material) return fail(RT_ERR_INVALID, "null object arrays");
  if (leaf_thr < 1) return fail(RT_ERR_INVALID, "leafThreshold must be >= 1 (the reference recurses forever on a 1-object list otherwise)");
  for (int i = 0; i < n; ++i)
    if (kind[i] < RT_PRIM_TRIANGLE || kind[i] > RT_PRIM_PLANE) return fail(RT_ERR_INVALID, "object %d has unknown kind %d", i, kind[i]);

  const bool dbg_t = getenv("RTH_DEBUG") != nullptr;      // RTH_DEBUG: where the build time goes
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!dbg_t) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "rthost: bvh_build %-24s %7.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  std::vector<Box> boxes((size_t)n);
  for (int i = 0; i < n; ++i) boxes[i] = prim_box(kind[i], data + 9 * (size_t)i);   // cachedObjs, BVH.hs:31

  const long total_nodes = subtree_nodes(n, 0, leaf_thr, max_depth);
  if (total_nodes > (1L << 30)) return fail(RT_ERR_UNSUPPORTED, "tree too large");
  auto *bvh = new rth_bvh();
  bvh->n_nodes = (int)total_nodes;
  bvh->node_bounds.assign(6 * (size_t)total_nodes, 0.0);
  bvh->node_child.assign(2 * (size_t)total_nodes, 0);
  bvh->prim_kind.resize(n); bvh->prim_material.resize(n); bvh->prim_id.resize(n); bvh->prim_data.resize(9 * (size_t)n);

  std::vector<int32_t> idx((size_t)n);
  for (int i = 0; i < n; ++i) idx[i] = i;
  lap("boxes + allocation");

  std::mutex mu;
  std::condition_variable cv;
  std::deque<Task> queue;
  std::atomic<long> pending(0);
  std::atomic<int> n_leaves(0), max_d(0);
  const long SPAWN_MIN = 1 << 15;
  const bool parallel = n_threads > 1;

  auto push_task = [&](const Task &t) {
    pending.fetch_add(1);
    { std::lock_guard<std::mutex> g(mu); queue.push_back(t); }
    cv.notify_one();
  };

  // processes one subtree; large children become tasks of their own when running in parallel
  std::vector<std::pair<double, int32_t>> scratch_main;
  auto process = [&](auto &&self, Task t, std::vector<std::pair<double, int32_t>> &keyed) -> void {
    const long count = t.e - t.b;
    double *nb = bvh->node_bounds.data() + 6 * t.node;
    if (count <= leaf_thr || t.depth >= max_depth) {                                  // BVH.hs:36
      Box bb = boxes[idx[t.b]];
      for (long i = t.b + 1; i < t.e; ++i) {
        const Box &o = boxes[idx[i]];
        for (int k = 0; k < 3; ++k) { bb.lo[k] = std::min(o.lo[k], bb.lo[k]); bb.hi[k] = std::max(o.hi[k], bb.hi[k]); }
      }
      for (int k = 0; k < 3; ++k) { nb[k] = bb.lo[k]; nb[3 + k] = bb.hi[k]; }
      bvh->node_child[2 * t.node] = (int32_t)(-1 - t.prim0);
      bvh->node_child[2 * t.node + 1] = (int32_t)count;
      for (long i = 0; i < count; ++i) {                                              // DV.fromList (map fst objs), :38
        int32_t src = idx[t.b + i];
        long dst = t.prim0 + i;
        bvh->prim_kind[dst] = kind[src]; bvh->prim_material[dst] = material[src]; bvh->prim_id[dst] = src;
        std::memcpy(bvh->prim_data.data() + 9 * dst, data + 9 * (size_t)src, 9 * sizeof(double));
      }
      n_leaves.fetch_add(1);
      int d = max_d.load();
      while (t.depth > d && !max_d.compare_exchange_weak(d, t.depth)) {}
      return;
    }
    // chooseSplitAxis, BVH.hs:51-71
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (long i = t.b; i < t.e; ++i) {
      const Box &o = boxes[idx[i]];
      for (int k = 0; k < 3; ++k) { mn[k] = std::min(o.lo[k], mn[k]); mx[k] = std::max(o.hi[k], mx[k]); }
    }
    const double ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    const int axis = (ex > ey && ex > ez) ? 0 : (ey > ez ? 1 : 2);
    // sortByCentroid (stable), :74-80
    keyed.resize((size_t)count);
    for (long i = 0; i < count; ++i) {
      const Box &o = boxes[idx[t.b + i]];
      keyed[i] = {0.5 * (o.lo[axis] + o.hi[axis]), idx[t.b + i]};
    }
    auto by_key = [](const std::pair<double, int32_t> &a, const std::pair<double, int32_t> &b) { return a.first < b.first; };
    if (parallel && count >= (1L << 18)) {
      // the top nodes are the serial part of the build: sort pieces on their own threads, then merge neighbours
      // (stable pieces merged in position order = the stable sort of the whole)
      int pieces = 1;
      while (pieces * 2 <= std::min(n_threads, 16)) pieces *= 2;
      std::vector<long> cut(pieces + 1);
      for (int k = 0; k <= pieces; ++k) cut[k] = count * k / pieces;
      {
        std::vector<std::future<void>> fs;
        for (int k = 0; k < pieces; ++k)
          fs.push_back(std::async(std::launch::async, [&, k]() { std::stable_sort(keyed.begin() + cut[k], keyed.begin() + cut[k + 1], by_key); }));
        for (auto &f : fs) f.get();
      }
      for (int w = 1; w < pieces; w *= 2) {
        std::vector<std::future<void>> fs;
        for (int k = 0; k + w < pieces; k += 2 * w)
          fs.push_back(std::async(std::launch::async, [&, k, w]() {
            std::inplace_merge(keyed.begin() + cut[k], keyed.begin() + cut[k + w], keyed.begin() + cut[std::min(pieces, k + 2 * w)], by_key);
          }));
        for (auto &f : fs) f.get();
      }
    } else {
      std::stable_sort(keyed.begin(), keyed.end(), by_key);
    }
    for (long i = 0; i < count; ++i) idx[t.b + i] = keyed[i].second;
    const long half = count / 2;                                                      // :42
    const long lnodes = subtree_nodes(half, t.depth + 1, leaf_thr, max_depth);
    Task L{t.b, t.b + half, t.depth + 1, t.node + 1, t.prim0};
    Task R{t.b + half, t.e, t.depth + 1, t.node + 1 + lnodes, t.prim0 + half};
    bvh->node_child[2 * t.node] = (int32_t)L.node;
    bvh->node_child[2 * t.node + 1] = (int32_t)R.node;
    if (parallel && count >= 2 * SPAWN_MIN) { push_task(L); push_task(R); }
    else { self(self, L, keyed); self(self, R, keyed); }
  };

  if (!parallel) {
    process(process, Task{0, n, 0, 0, 0}, scratch_main);
  } else {
    push_task(Task{0, n, 0, 0, 0});
    auto worker = [&]() {
      std::vector<std::pair<double, int32_t>> keyed;
      for (;;) {
        Task t;
        {
          std::unique_lock<std::mutex> g(mu);
          cv.wait(g, [&] { return !queue.empty() || pending.load() == 0; });
          if (queue.empty()) return;
          t = queue.front(); queue.pop_front();
        }
        process(process, t, keyed);
        if (pending.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(mu); cv.notify_all(); }
      }
    };
    std::vector<std::thread> th;
    for (int i = 0; i < n_threads; ++i) th.emplace_back(worker);
    for (auto &x : th) x.join();
  }
  lap("sorts + leaves");
  // internal boxes bottom-up: in pre-order a child's index is larger than its parent's (BVH.hs:45)
  for (long i = total_nodes - 1; i >= 0; --i) {
    if (bvh->node_child[2 * i] < 0) continue;
    const double *l = bvh->node_bounds.data() + 6 * (size_t)bvh->node_child[2 * i];
    const double *r = bvh->node_bounds.data() + 6 * (size_t)bvh->node_child[2 * i + 1];
    double *nb = bvh->node_bounds.data() + 6 * i;
    for (int k = 0; k < 3; ++k) { nb[k] = std::min(r[k], l[k]); nb[3 + k] = std::max(r[3 + k], l[3 + k]); }
  }
  lap("internal boxes");
  bvh->n_leaves = n_leaves.load();
  bvh->depth = max_d.load();
  *out = bvh;
  return RT_OK;
}

void rth_bvh_destroy(rth_bvh *b) { delete b; }

void rth_bvh_info(const rth_bvh *b, int32_t *n_nodes, int32_t *n_leaves, int32_t *depth, int32_t *n_prims) {
  if (n_nodes) *n_nodes = b->n_nodes;
  if (n_leaves) *n_leaves = b->n_leaves;
  if (depth) *depth = b->depth;
  if (n_prims) *n_prims = (int32_t)b->prim_kind.size();
}

int rth_bvh_fill_desc(const rth_bvh *b, rt_scene_desc *d) {
  if (!b || !d) return fail(RT_ERR_INVALID, "null argument");
  d->n_nodes = b->n_nodes;
  d->node_bounds = b->node_bounds.data();
  d->node_child = b->node_child.data();
  d->n_prims = (int32_t)b->prim_kind.size();
  d->prim_kind = b->prim_kind.data();
  d->prim_data = b->prim_data.data();
  d->prim_material = b->prim_material.data();
  d->prim_id = b->prim_id.data();
  return RT_OK;
}

// defaultCamera, Rendering/Camera.hs:26-44
int rth_default_camera(const double lf[3], const double la[3], const double up[3], double vfov, double aspect,
                       double aperture, double focus, rt_camera *c) {
  if (!lf || !la || !up || !c) return fail(RT_ERR_INVALID, "null argument");
  auto norm = [](double *v) { double s = 1 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] = s * v[0]; v[1] = s * v[1]; v[2] = s * v[2]; };
  auto cross = [](const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
  };
  const double theta = vfov * M_PI / 180.0;            // degreesToRadians, Utils/Constants.hs:19-20
  const double h = std::tan(theta / 2);                // :29
  const double vh = 2.0 * h * focus;                   // :30
  const double vw = vh * aspect;                       // :31
  double w[3] = {lf[0] - la[0], lf[1] - la[1], lf[2] - la[2]};
  norm(w);                                             // :33
  double u[3]; cross(up, w, u); norm(u);               // :34
  double v[3]; cross(w, u, v);                         // :35
  for (int k = 0; k < 3; ++k) {
    c->origin[k] = lf[k];
    c->horizontal[k] = vw * u[k];                      // :37
    c->vertical[k] = vh * v[k];                        // :38
    double t1 = lf[k] - 0.5 * c->horizontal[k];        // :40
    double t2 = t1 - 0.5 * c->vertical[k];             // :41
    c->lower_left[k] = t2 - focus * w[k];              // :42
    c->u[k] = u[k]; c->v[k] = v[k]; c->w[k] = w[k];
  }
  c->aperture = aperture;
  return RT_OK;
}

}  // extern "C"
