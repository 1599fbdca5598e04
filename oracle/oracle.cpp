// oracle.cpp — CPU FP64 restatement of omeldar/raytracer's per-pixel ray<->scene hot path.
//
// THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may load it.  The product (raytracer_amd/csrc) never
// links, includes or calls anything in this directory.
//
// PARITY UNPINNED: the reference is Haskell, no ghc/cabal/stack exists in the build
// container, the reference has no test-suite, no golden vectors and no
// fixture scenes (raytracer.cabal defines two executables only; .gitignore:28-33 drops
// all assets).  Fidelity therefore rests on following the cited source lines literally
// and on the hand-derived known-answer tests in tests/test_oracle_kat.py; a second, separately
// written restatement (oracle/pyref.py) must agree with this one bit for bit
// (tests/test_oracle_pyref.py).
//
// Every function names the reference lines it restates (paths relative to
// /root/reference/app).  Arithmetic is binary64 with the reference's operation order;
// build with -ffp-contract=off so no FMA is formed that GHC would not form.
//
// The ONLY intentional departure: the reference seeds its RNGs from OS entropy
// (Utils/RandomHelper.hs:22, RenderWorker.hs:103-105) and is not reproducible.  The
// oracle draws the same distributions from Philox4x32-10 keyed by
// (seed; pixel, sample, stream, sub) — DESIGN.md §RNG — so GPU and oracle can be
// compared path-for-path.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "../include/rtcore.h"

namespace {

// ---------------------------------------------------------------- Core/Vec3.hs:33-153
struct V3 { double x, y, z; };
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }              // :54-55
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }              // :58-59
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // :62-63
inline V3 cross(V3 a, V3 b) {                                                        // :66-67
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline V3 scale(double s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }              // :70-71
inline double vlength(V3 v) { return std::sqrt(dot(v, v)); }                         // :74-75
inline V3 normalize(V3 v) { return scale(1 / vlength(v), v); }                       // :78-79
inline V3 negateV(V3 v) { return {-v.x, -v.y, -v.z}; }                               // :82-83
inline V3 mul(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }              // :86-87
inline V3 reflect(V3 v, V3 n) { return sub(v, scale(2 * dot(v, n), n)); }            // :98-99
inline V3 refract(V3 v, V3 n, double eta) {                                          // :102-107
  double cosTheta = -dot(v, n);
  V3 rOutPerp = scale(eta, add(v, scale(cosTheta, n)));
  V3 rOutParallel = scale(-std::sqrt(std::fabs(1.0 - dot(rOutPerp, rOutPerp))), n);
  return add(rOutPerp, rOutParallel);
}
inline double lengthSquared(V3 v) { return dot(v, v); }                              // :110-111
inline V3 minV(V3 a, V3 b) { return {std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z)}; }  // :140-141
inline V3 maxV(V3 a, V3 b) { return {std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z)}; }  // :136-137

const double kInf = std::numeric_limits<double>::infinity();                         // Utils/Constants.hs:9-10
const double kEps = 1e-8;                                                            // Utils/Constants.hs:15-16

// ---------------------------------------------------------------- Core/Ray.hs:11-19
struct Ray { V3 o, d; };
inline V3 at(const Ray &r, double t) { return add(r.o, scale(t, r.d)); }

// ---------------------------------------------------------------- Utils/Interval.hs:6-21
struct Interval { double lo, hi; };
inline bool contains(Interval i, double x) { return x >= i.lo && x <= i.hi; }

// ---------------------------------------------------------------- Hittable/BoundingBox.hs:8-37
struct AABB { V3 lo, hi; };
inline AABB surroundingBox(AABB a, AABB b) { return {minV(a.lo, b.lo), maxV(a.hi, b.hi)}; }  // :14-16

// hitAABBInterval, BoundingBox.hs:18-37.  Returns false for Nothing.
inline bool checkAxis(double o, double d, double minB, double maxB, double &a0, double &a1) {
  if (std::fabs(d) < 1e-8) {                       // :21
    if (o < minB || o > maxB) return false;        // :22
    a0 = -kInf; a1 = kInf;
    return true;
  }
  double invD = 1.0 / d;                           // :24
  double t0 = (minB - o) * invD;                   // :25
  double t1 = (maxB - o) * invD;                   // :26
  a0 = std::min(t0, t1); a1 = std::max(t0, t1);    // :27
  return true;
}
inline bool hitAABBInterval(const AABB &b, const Ray &r, Interval in, Interval &out) {
  double a0, a1, b0, b1, c0, c1;
  bool okx = checkAxis(r.o.x, r.d.x, b.lo.x, b.hi.x, a0, a1);
  bool oky = checkAxis(r.o.y, r.d.y, b.lo.y, b.hi.y, b0, b1);
  bool okz = checkAxis(r.o.z, r.d.z, b.lo.z, b.hi.z, c0, c1);
  if (!(okx && oky && okz)) return false;                         // :33
  double tEnter = std::max(std::max(a0, b0), std::max(c0, in.lo)); // :30 maximum [a0,b0,c0,tMin]
  double tExit = std::min(std::min(a1, b1), std::min(c1, in.hi));  // :31
  if (tEnter <= tExit) { out = {tEnter, tExit}; return true; }     // :32
  return false;
}

// ---------------------------------------------------------------- Hittable/Class.hs:8-26
struct HitRecord {
  V3 point, normal;
  double t;
  bool frontFace;
  int materialId;
  int primId;   // not in the reference: index in allObjects, reported for the ID-parity check
};
inline void setFaceNormal(const Ray &r, V3 outward, V3 &faceN, bool &front) {   // Class.hs:22-26
  front = dot(r.d, outward) < 0;
  faceN = front ? outward : negateV(outward);
}

// ---------------------------------------------------------------- primitives
struct Prim {
  int kind;       // RT_PRIM_*
  double d[9];
  int materialId;
  int id;
};

struct Counters {
  uint64_t rays_closest = 0, rays_shadow = 0;
  uint64_t nodes_literal = 0;     // every BVHInternal `hit` invocation of the reference
  uint64_t nodes_nonempty = 0;    // the subset entered with a non-empty interval (what a culling traversal fetches)
  uint64_t tri_tests = 0, sphere_tests = 0, plane_tests = 0;
  uint64_t tri_tests_nonempty = 0, sphere_tests_nonempty = 0, plane_tests_nonempty = 0;
  uint64_t shaded_hits = 0;
  void operator+=(const Counters &o) {
    rays_closest += o.rays_closest; rays_shadow += o.rays_shadow;
    nodes_literal += o.nodes_literal; nodes_nonempty += o.nodes_nonempty;
    tri_tests += o.tri_tests; sphere_tests += o.sphere_tests; plane_tests += o.plane_tests;
    tri_tests_nonempty += o.tri_tests_nonempty; sphere_tests_nonempty += o.sphere_tests_nonempty;
    plane_tests_nonempty += o.plane_tests_nonempty; shaded_hits += o.shaded_hits;
  }
};

// Triangle: Hittable/Objects/Triangle.hs:23-64
inline AABB boxTriangle(const Prim &p) {                                     // :24-28
  V3 a{p.d[0], p.d[1], p.d[2]}, b{p.d[3], p.d[4], p.d[5]}, c{p.d[6], p.d[7], p.d[8]};
  V3 mn = minV(minV(a, b), c), mx = maxV(maxV(a, b), c);
  V3 pad{kEps, kEps, kEps};
  return {sub(mn, pad), add(mx, pad)};
}
inline bool hitTriangle(const Prim &p, const Ray &ray, Interval in, HitRecord &rec) {   // :30-64
  V3 p0{p.d[0], p.d[1], p.d[2]}, p1{p.d[3], p.d[4], p.d[5]}, p2{p.d[6], p.d[7], p.d[8]};
  const double eps = 1e-8;                        // :31
  V3 e1 = sub(p1, p0);                            // :32
  V3 e2 = sub(p2, p0);                            // :33
  V3 h = cross(ray.d, e2);                        // :34
  double a = dot(e1, h);                          // :35
  if (std::fabs(a) < eps) return false;           // :36-37
  double f = 1.0 / a;                             // :39
  V3 s = sub(ray.o, p0);                          // :40
  double u = f * dot(s, h);                       // :41
  if (u < 0.0 || u > 1.0) return false;           // :42-43
  V3 q = cross(s, e1);                            // :45
  double v = f * dot(ray.d, q);                   // :46
  if (v < 0.0 || u + v > 1.0) return false;       // :47-48
  double t = f * dot(e2, q);                      // :50
  V3 normalV = normalize(cross(e1, e2));          // :51
  V3 fn; bool front;
  setFaceNormal(ray, normalV, fn, front);         // :52
  if (!contains(in, t)) return false;             // :53,64
  rec.point = at(ray, t); rec.normal = fn; rec.t = t; rec.frontFace = front;   // :56-62
  rec.materialId = p.materialId; rec.primId = p.id;
  return true;
}

// Sphere: Hittable/Objects/Sphere.hs:19-40 (near root only)
inline AABB boxSphere(const Prim &p) {                                       // :20-22
  V3 c{p.d[0], p.d[1], p.d[2]}; double r = p.d[3];
  V3 rv{r, r, r};
  return {sub(c, rv), add(c, rv)};
}
inline bool hitSphere(const Prim &p, const Ray &ray, Interval in, HitRecord &rec) {     // :24-40
  V3 c{p.d[0], p.d[1], p.d[2]}; double r = p.d[3];
  V3 oc = sub(ray.o, c);                                          // :29
  double a = lengthSquared(ray.d);                                // :30
  double h = dot(oc, ray.d);                                      // :31
  double disc = h * h - a * (lengthSquared(oc) - r * r);          // :32
  if (disc < 0) return false;                                     // :25
  double sqrtD = std::sqrt(disc);                                 // :33
  double t1 = (-h - sqrtD) / a;                                   // :34
  if (!contains(in, t1)) return false;                            // :26-27
  V3 pt = at(ray, t1);                                            // :37
  V3 outward = scale(1.0 / r, sub(pt, c));                        // :38
  V3 fn; bool front;
  setFaceNormal(ray, outward, fn, front);                         // :39
  rec.point = pt; rec.normal = fn; rec.t = t1; rec.frontFace = front;   // :40
  rec.materialId = p.materialId; rec.primId = p.id;
  return true;
}

// Plane: Hittable/Objects/Plane.hs:19-32 (note the constant, position-independent box)
inline AABB boxPlane(const Prim &) { return {V3{-1e6, -1e-3, -1e6}, V3{1e6, 1e-3, 1e6}}; }   // :20
inline bool hitPlane(const Prim &p, const Ray &ray, Interval in, HitRecord &rec) {      // :21-32
  V3 pp{p.d[0], p.d[1], p.d[2]}, n{p.d[3], p.d[4], p.d[5]};
  double denom = dot(ray.d, n);                                   // :22
  if (!(std::fabs(denom) > 1e-6)) return false;                   // :23,32
  double t = dot(sub(pp, ray.o), n) / denom;                      // :25
  if (!contains(in, t)) return false;                             // :26,31
  V3 fn; bool front;
  setFaceNormal(ray, n, fn, front);                               // :29
  rec.point = at(ray, t); rec.normal = fn; rec.t = t; rec.frontFace = front;   // :28-30
  rec.materialId = p.materialId; rec.primId = p.id;
  return true;
}

inline AABB boxOf(const Prim &p) {             // Hittable/Helpers.hs getBox -> boundingBox
  switch (p.kind) {
    case RT_PRIM_TRIANGLE: return boxTriangle(p);
    case RT_PRIM_SPHERE: return boxSphere(p);
    default: return boxPlane(p);
  }
}
inline bool hitPrim(const Prim &p, const Ray &r, Interval in, HitRecord &rec, Counters *c) {
  bool nonempty = in.lo <= in.hi;
  switch (p.kind) {
    case RT_PRIM_TRIANGLE: if (c) { c->tri_tests++; c->tri_tests_nonempty += nonempty; } return hitTriangle(p, r, in, rec);
    case RT_PRIM_SPHERE: if (c) { c->sphere_tests++; c->sphere_tests_nonempty += nonempty; } return hitSphere(p, r, in, rec);
    default: if (c) { c->plane_tests++; c->plane_tests_nonempty += nonempty; } return hitPlane(p, r, in, rec);
  }
}

// ---------------------------------------------------------------- Hittable/BVH.hs
struct BVHNode {                     // :20-22
  bool leaf;
  AABB box;
  std::vector<Prim> objs;            // BVHLeaf
  std::unique_ptr<BVHNode> left, right;   // BVHInternal
};

struct ObjBox { Prim obj; AABB box; };

// chooseSplitAxis, BVH.hs:51-71
int chooseSplitAxis(const std::vector<ObjBox> &objs) {
  V3 vmin{kInf, kInf, kInf}, vmax{-kInf, -kInf, -kInf};
  for (size_t i = objs.size(); i-- > 0;) {            // foldr: from the right; min/max are exact so order is immaterial
    vmin = {std::min(objs[i].box.lo.x, vmin.x), std::min(objs[i].box.lo.y, vmin.y), std::min(objs[i].box.lo.z, vmin.z)};
    vmax = {std::max(objs[i].box.hi.x, vmax.x), std::max(objs[i].box.hi.y, vmax.y), std::max(objs[i].box.hi.z, vmax.z)};
  }
  V3 extent = sub(vmax, vmin);                                             // :68
  if (extent.x > extent.y && extent.x > extent.z) return 0;                // :69-70
  return extent.y > extent.z ? 1 : 2;                                      // :71
}
inline double axisValue(int axis, V3 v) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; }   // :82-86

// buildBVH, BVH.hs:34-48 (+ sortByCentroid :74-80, surroundingBoxList :101-103)
std::unique_ptr<BVHNode> buildBVH(std::vector<ObjBox> objs, int depth, int leafThreshold, int maxDepth) {
  auto node = std::make_unique<BVHNode>();
  long objCount = (long)objs.size();                                       // :48
  if (objCount <= leafThreshold || depth >= maxDepth) {                    // :36
    AABB bb = objs[0].box;                                                 // :37 surroundingBoxList snd objs
    for (size_t i = 1; i < objs.size(); ++i) bb = surroundingBox(objs[i].box, bb);
    node->leaf = true; node->box = bb;
    node->objs.reserve(objs.size());
    for (auto &ob : objs) node->objs.push_back(ob.obj);                    // :38
    return node;
  }
  int axis = chooseSplitAxis(objs);                                        // :40
  // List.sortOn (stable merge sort on the centroid key), :74-80
  std::vector<std::pair<double, size_t>> keyed(objs.size());
  for (size_t i = 0; i < objs.size(); ++i) {
    V3 centroid = scale(0.5, add(objs[i].box.lo, objs[i].box.hi));         // :78
    keyed[i] = {axisValue(axis, centroid), i};
  }
  std::stable_sort(keyed.begin(), keyed.end(),
                   [](const std::pair<double, size_t> &a, const std::pair<double, size_t> &b) { return a.first < b.first; });
  size_t half = (size_t)(objCount / 2);                                    // :42 splitAt (objCount `div` 2)
  std::vector<ObjBox> l, r;
  l.reserve(half); r.reserve(objs.size() - half);
  for (size_t i = 0; i < keyed.size(); ++i) (i < half ? l : r).push_back(objs[keyed[i].second]);
  objs.clear(); objs.shrink_to_fit();
  node->leaf = false;
  node->left = buildBVH(std::move(l), depth + 1, leafThreshold, maxDepth);   // :43
  node->right = buildBVH(std::move(r), depth + 1, leafThreshold, maxDepth);  // :44
  node->box = surroundingBox(node->right->box, node->left->box);             // :45 foldr over [left,right] seeded with left
  return node;
}

// instance Hittable BVHNode, BVH.hs:120-146; closestHitList, Utils/HitHelpers.hs:11-16
bool hitBVH(const BVHNode *n, const Ray &ray, Interval in, HitRecord &out, Counters *c) {
  if (n->leaf) {                                                             // :121-122
    // foldr closer Nothing: walk from the right; `closer (Just x) (Just y) = if t x < t y then x else y`
    bool have = false; HitRecord acc{};
    for (size_t i = n->objs.size(); i-- > 0;) {
      HitRecord x;
      if (!hitPrim(n->objs[i], ray, in, x, c)) continue;                     // closer Nothing acc = acc
      if (!have) { acc = x; have = true; }                                   // closer (Just x) Nothing
      else if (x.t < acc.t) acc = x;                                         // else keep y (later element wins ties)
    }
    if (have) out = acc;
    return have;
  }
  if (c) { c->nodes_literal++; c->nodes_nonempty += (in.lo <= in.hi); }
  Interval lInt, rInt;
  bool hl = hitAABBInterval(n->left->box, ray, in, lInt);                    // :124
  bool hr = hitAABBInterval(n->right->box, ray, in, rInt);
  if (!hl && !hr) return false;                                              // :125
  if (hl && !hr) return hitBVH(n->left.get(), ray, lInt, out, c);            // :126
  if (!hl && hr) return hitBVH(n->right.get(), ray, rInt, out, c);           // :127
  const BVHNode *first, *second; Interval iFirst, iSecond;                   // :128-134
  if (lInt.lo < rInt.lo) { first = n->left.get(); iFirst = lInt; second = n->right.get(); iSecond = rInt; }
  else { first = n->right.get(); iFirst = rInt; second = n->left.get(); iSecond = lInt; }
  HitRecord h1, h2;
  bool got1 = hitBVH(first, ray, iFirst, h1, c);                             // :136
  double newMaxT = got1 ? h1.t : iSecond.hi;                                 // :137
  Interval tighter{iSecond.lo, newMaxT};                                     // :138
  bool got2 = hitBVH(second, ray, tighter, h2, c);                           // :139
  if (got2) {
    if (got1) out = (h1.t < h2.t) ? h1 : h2;                                 // :141
    else out = h2;                                                           // :142
    return true;
  }
  if (got1) { out = h1; return true; }                                       // :143
  return false;
}

// ---------------------------------------------------------------- scene
struct Material {      // Rendering/Material.hs:13-21 (fields the integrator reads)
  V3 diffuse{1, 1, 1};
  bool hasEmission = false; V3 emission{0, 0, 0};
  bool hasShininess = false; double shininess = 0;
  bool hasIor = false; double ior = 0;
  bool hasTransmission = false; double transmission = 0;
};
struct Light { int kind; V3 vec, intensity; };   // Rendering/Light.hs:12-14

struct Sky { int format = RT_SKY_NONE, w = 0, h = 0; std::vector<uint8_t> u8; std::vector<float> f32; };

struct Scene {
  std::unique_ptr<BVHNode> bvh;
  std::vector<std::pair<int, Material>> materials;   // Data.Map Int Material
  std::vector<Light> lights;
  int bgKind; V3 bg1, bg2;
  Sky sky;
  const Material *lookup(int id) const {             // M.lookup matId materialMap, Trace.hs:28
    for (auto &m : materials) if (m.first == id) return &m.second;
    return nullptr;
  }
};
const Material kDefaultMaterial{};                   // Rendering/Material.hs:25-35

// Rendering/ImageGenerator.hs:130-135 + Rendering/Color.hs:16-17
inline V3 lerp(double t, V3 start, V3 end) { return add(scale(1.0 - t, start), scale(t, end)); }
inline V3 backgroundColor(const Scene &s, const Ray &ray) {
  if (s.bgKind == RT_BG_SOLID) return s.bg1;                       // :131
  V3 unitDir = normalize(ray.d);                                   // :133
  double tval = 0.5 * (unitDir.y + 1.0);                           // :134
  return lerp(tval, s.bg2, s.bg1);                                 // :135  Col.lerp tval c2 c1
}

// atan2 on Double under GHC (SkySphere.hs:35 calls it): instance RealFloat Double does not define atan2, so the class DEFAULT of
// GHC.Float runs - the Haskell 2010 report's Prelude definition, restated guard by guard - and libm's atan2 is never called.
// It differs from libm's in the last bits (the division, atan and `pi + ...` each round), which decides the texel of a direction
// within rounding of a texel boundary.  atan on Double is libm's atan.
inline bool isNegativeZero(double v) { return v == 0.0 && std::signbit(v); }
inline double hsAtan2(double y, double x) {
  if (x > 0) return std::atan(y / x);
  if (x == 0 && y > 0) return M_PI / 2;
  if (x < 0 && y > 0) return M_PI + std::atan(y / x);
  if ((x <= 0 && y < 0) || (x < 0 && isNegativeZero(y)) || (isNegativeZero(x) && isNegativeZero(y))) return -hsAtan2(-y, x);
  if (y == 0 && (x < 0 || isNegativeZero(x))) return M_PI;   // must be after the previous test on zero y
  if (x == 0 && y == 0) return y;                            // must be after the other double zero tests
  return x + y;                                              // x or y is a NaN
}

// sampleSkySphere, Rendering/SkySphere.hs:32-50
inline int clampInt(long v, int lo, int hi) { return (int)std::max((long)lo, std::min(v, (long)hi)); }
inline V3 sampleSkySphere(const Sky &sky, V3 dir) {
  V3 d = normalize(dir);                                                     // :34
  double dy = std::max(-1.0, std::min(1.0, d.y));   // guard: reference would take asin of 1+ulp (NaN -> garbage index)
  double u = 0.5 + hsAtan2(d.z, d.x) / (2 * M_PI);                           // :35
  double v = 0.5 - std::asin(dy) / M_PI;                                     // :36
  int i = clampInt((long)std::floor(u * (double)sky.w), 0, sky.w - 1);       // :39
  int j = clampInt((long)std::floor(v * (double)sky.h), 0, sky.h - 1);       // :40
  size_t k = ((size_t)j * sky.w + i) * 3;                                    // pixelAt img i j
  if (sky.format == RT_SKY_RGB8)                                             // :42-44
    return {sky.u8[k] / 255.0, sky.u8[k + 1] / 255.0, sky.u8[k + 2] / 255.0};
  return {(double)sky.f32[k], (double)sky.f32[k + 1], (double)sky.f32[k + 2]};   // :45-47
}

// ---------------------------------------------------------------- RNG
inline void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
inline double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }        // U(0,1)
inline double um11(uint32_t w) { return ((double)w + 0.5) * (1.0 / 2147483648.0) - 1.0; } // U(-1,1)

struct RngKey { uint32_t key[2]; uint32_t pixel, sample; };
inline void draw(const RngKey &k, uint32_t stream, uint32_t sub, uint32_t out[4]) {
  uint32_t ctr[4] = {k.pixel, k.sample, stream, sub};
  philox4x32_10(ctr, k.key, out);
}

// ---------------------------------------------------------------- Rendering/Camera.hs
// defaultCamera, Camera.hs:26-44
rt_camera defaultCamera(V3 lookFrom, V3 lookAt, V3 vUp, double vfov, double aspect, double aperture, double focusDist) {
  double theta = vfov * M_PI / 180.0;                 // Utils/Constants.hs:19-20
  double h = std::tan(theta / 2);                     // :29
  double viewportHeight = 2.0 * h * focusDist;        // :30
  double viewportWidth = viewportHeight * aspect;     // :31
  V3 camW = normalize(sub(lookFrom, lookAt));         // :33
  V3 camU = normalize(cross(vUp, camW));              // :34
  V3 camV = cross(camW, camU);                        // :35
  V3 hor = scale(viewportWidth, camU);                // :37
  V3 ver = scale(viewportHeight, camV);               // :38
  V3 t1 = sub(lookFrom, scale(0.5, hor));             // :40
  V3 t2 = sub(t1, scale(0.5, ver));                   // :41
  V3 llc = sub(t2, scale(focusDist, camW));           // :42
  rt_camera c;
  auto put = [](double *dst, V3 v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; };
  put(c.origin, lookFrom); put(c.lower_left, llc); put(c.horizontal, hor); put(c.vertical, ver);
  put(c.u, camU); put(c.v, camV); put(c.w, camW); c.aperture = aperture;
  return c;
}
inline V3 ld(const double *p) { return {p[0], p[1], p[2]}; }

// generateRay, Camera.hs:47-60.  (i, j) as the reference passes them: j = rowIdx counted from the bottom.
// Random inputs: uOffset, vOffset and the unit-disk lens sample come from stream 0 of the pixel/sample key.
inline Ray generateRay(const rt_camera &cam, int i, int j, int width, int height, bool antialiasing, const RngKey &key) {
  uint32_t w[4];
  draw(key, 0, 0, w);
  double uOffset = antialiasing ? u01(w[0]) : 0.5;        // RenderWorker.hs:98
  double vOffset = antialiasing ? u01(w[1]) : 0.5;        // RenderWorker.hs:99
  // randomInUnitDisk, Core/Vec3.hs:126-133 (rejection).  aperture == 0 multiplies the sample by 0,
  // so the loop is skipped then (result-identical).
  double lx = 0, ly = 0;
  if (cam.aperture != 0.0) {
    lx = um11(w[2]); ly = um11(w[3]);
    uint32_t sub = 1;
    while (!(lx * lx + ly * ly < 1.0) && sub < 64) {
      draw(key, 0, sub, w);
      lx = um11(w[0]); ly = um11(w[1]);
      if (!(lx * lx + ly * ly < 1.0)) { lx = um11(w[2]); ly = um11(w[3]); }
      ++sub;
    }
    if (!(lx * lx + ly * ly < 1.0)) { lx = 0; ly = 0; }
  }
  V3 lensOffset = scale(cam.aperture / 2, add(scale(lx, ld(cam.u)), scale(ly, ld(cam.v))));   // :50
  V3 origin2 = add(ld(cam.origin), lensOffset);                                               // :51
  double ww = (double)width, hh = (double)height;                                             // :52-53
  double rayu = ((double)i + uOffset) / ww;                                                   // :54
  double rayv = ((double)j + vOffset) / hh;                                                   // :55
  V3 direction = sub(add(add(ld(cam.lower_left), scale(rayu, ld(cam.horizontal))), scale(rayv, ld(cam.vertical))), origin2);  // :56-59
  return {origin2, normalize(direction)};                                                     // :60
}

// ---------------------------------------------------------------- Rendering/Light.hs:16-49
V3 checkLightVisibility(const Scene &s, V3 origin, V3 lnormal, const Light &light, Counters *c) {
  if (light.kind == RT_LIGHT_POINT) {                                        // :26-38
    V3 toLight = normalize(sub(light.vec, origin));                          // :27
    double dist = vlength(sub(light.vec, origin));                           // :28
    Ray shadowRay{origin, toLight};                                          // :29
    HitRecord tmp;
    if (c) c->rays_shadow++;
    bool blocked = hitBVH(s.bvh.get(), shadowRay, {0.001, dist - 0.01}, tmp, c);   // :30
    if (blocked) return {0, 0, 0};                                           // :33-34
    double lightPower = std::max(0.0, dot(lnormal, toLight));                // :36
    double attenuation = 1 / (dist * dist);                                  // :37
    return scale(lightPower * attenuation, light.intensity);                 // :38
  }
  V3 toLight = normalize(scale(-1, light.vec));                              // :40
  Ray shadowRay{origin, toLight};                                            // :41
  HitRecord tmp;
  if (c) c->rays_shadow++;
  bool blocked = hitBVH(s.bvh.get(), shadowRay, {0.001, 10000}, tmp, c);     // :42
  if (blocked) return {0, 0, 0};                                             // :45-46
  double lightPower = std::max(0.0, dot(lnormal, toLight));                  // :48
  return scale(lightPower, light.intensity);                                 // :49
}
V3 computeLighting(const Scene &s, const HitRecord &rec, Counters *c) {      // :16-21
  V3 hitPoint = rec.point;
  V3 hitNormal = normalize(rec.normal);                                      // :19
  V3 acc{0, 0, 0};
  for (size_t i = s.lights.size(); i-- > 0;)                                 // :21 foldr V.add 0
    acc = add(checkLightVisibility(s, hitPoint, hitNormal, s.lights[i], c), acc);
  return acc;
}

// ---------------------------------------------------------------- Rendering/Trace.hs
inline double schlick(double cosine, double refIdx) {                        // :85-89
  double r0 = (1 - refIdx) / (1 + refIdx);
  double r0sq = r0 * r0;
  return r0sq + (1 - r0sq) * std::pow(1 - cosine, 5);
}
// GHC's Ord Double defaults (GHC.Classes): max x y = if x <= y then y else x; min x y = if x <= y then x else y.  On a NaN they are
// not std::max / std::min / fmax / fmin: min hi NaN = NaN, max lo NaN = lo - the reference's clamp turns a NaN bounce term into lo.
inline double hsMin(double x, double y) { return x <= y ? x : y; }
inline double hsMax(double x, double y) { return x <= y ? y : x; }
inline V3 clampV(V3 c, double lo, double hi) {                               // :91-94  cl v = max lo (min hi v)
  auto cl = [&](double v) { return hsMax(lo, hsMin(hi, v)); };
  return {cl(c.x), cl(c.y), cl(c.z)};
}
inline V3 toneMap(double e, V3 c) {                                          // :96-101
  return {c.x * e / (c.x * e + 1), c.y * e / (c.y * e + 1), c.z * e / (c.z * e + 1)};
}
inline V3 gammaCorrect(double g, V3 c) { return {std::pow(c.x, g), std::pow(c.y, g), std::pow(c.z, g)}; }   // :103-105

inline V3 backgroundSample(const Scene &s, V3 attenuation, const Ray &ray) {   // :78-83
  if (s.sky.format != RT_SKY_NONE) return mul(attenuation, sampleSkySphere(s.sky, ray.d));
  return mul(attenuation, backgroundColor(s, ray));
}

// traceLoop, Trace.hs:22-76.  `bounce` counts hits so far (selects the RNG stream that replaces
// the threaded StdGen: the 4 draws of :38-41 come from stream 1+bounce).
//
// rrStart > 0 is the EXTENSION BASELINE.json's config 4 names ("Russian roulette"; not in the reference: Config.hs:57
// "Russian roulette removed"), restated here only so that the kernel's opt-in implementation has a checker; with
// rrStart = 0 the function below is the reference's traceLoop, untouched.  Definition: once a path
// has made rrStart bounces, each further bounce survives with probability q = clamp(max channel of the new
// attenuation, 0.05, 1), decided by a uniform drawn from the bounce's RNG stream (sub-counter 1); a survivor's deeper
// value is weighted 1/q OUTSIDE the reference's clamp - emitted + lit + (1/q) clamp(bounce, 0, 10) - and a terminated
// path adds nothing deeper.  Unbiased for the reference's recursion when the clamps are inactive.
V3 traceLoop(const Scene &s, const Ray &ray0, const Ray &ray, int depth, V3 attenuation, const RngKey &key, int bounce, Counters *c, int rrStart = 0) {
  if (depth == 0) return backgroundSample(s, attenuation, ray0);             // :22
  HitRecord rec;
  if (c) c->rays_closest++;
  if (!hitBVH(s.bvh.get(), ray, {0.001, 10000}, rec, c))                     // :24
    return backgroundSample(s, attenuation, ray);                            // :25
  if (c) c->shaded_hits++;
  const Material *mp = s.lookup(rec.materialId);                             // :27-28
  const Material &mat = mp ? *mp : kDefaultMaterial;
  V3 emitted = mat.hasEmission ? mat.emission : V3{0, 0, 0};                 // :29
  V3 surfaceColor = mat.diffuse;                                             // :30
  V3 outwardNormal = rec.normal;                                             // :31
  V3 unitDir = normalize(ray.d);                                             // :32
  V3 hitPoint = rec.point;                                                   // :33
  V3 directLight = computeLighting(s, rec, c);                               // :35
  V3 litColor = mul(surfaceColor, directLight);                              // :36
  uint32_t w[4];
  draw(key, 1u + (uint32_t)bounce, 0, w);
  double randX = um11(w[0]), randY = um11(w[1]), randZ = um11(w[2]);         // :38-40
  double randD = u01(w[3]);                                                  // :41
  V3 randVec = normalize(V3{randX, randY, randZ});                           // :42
  bool tfrontFace = rec.frontFace;                                           // :44
  bool transIsOne = mat.hasTransmission && mat.transmission == 1.0;
  Ray nextRay;
  if (transIsOne && mat.hasIor) {                                            // :48-59
    double refIdx = mat.ior;
    double eta = tfrontFace ? 1.0 / refIdx : refIdx;                         // :50
    double cosTheta = hsMin(-dot(unitDir, outwardNormal), 1.0);              // :51 (GHC's min: 1.0 on a NaN)
    double sinTheta = std::sqrt(1.0 - cosTheta * cosTheta);                  // :52
    bool cannotRefract = eta * sinTheta > 1.0;                               // :53
    double reflectProb = schlick(cosTheta, eta);                             // :54
    V3 bouncedir;
    if (cannotRefract) bouncedir = reflect(unitDir, outwardNormal);          // :56
    else if (randD < reflectProb) bouncedir = reflect(unitDir, outwardNormal);   // :57
    else bouncedir = refract(unitDir, outwardNormal, eta);                   // :58
    nextRay = {hitPoint, bouncedir};                                         // :59
  } else if (mat.hasShininess && mat.shininess > 100) {                      // :60-64
    V3 reflected = reflect(unitDir, outwardNormal);
    V3 fuzzed = add(reflected, scale(0.05, randVec));
    nextRay = {hitPoint, normalize(fuzzed)};
  } else {                                                                   // :65-67
    V3 scatterDir = add(outwardNormal, randVec);
    nextRay = {hitPoint, normalize(scatterDir)};
  }
  V3 newAttenuation = transIsOne ? attenuation : mul(attenuation, surfaceColor);   // :69-72
  double rrWeight = 1.0;
  if (rrStart > 0 && bounce + 1 >= rrStart && depth - 1 > 0) {               // (extension) roulette before the next bounce is traced
    double q = std::min(1.0, std::max(0.05, std::max(newAttenuation.x, std::max(newAttenuation.y, newAttenuation.z))));
    uint32_t w2[4];
    draw(key, 1u + (uint32_t)bounce, 1, w2);
    if (!(u01(w2[0]) < q)) return add(emitted, litColor);                    // terminated: nothing deeper
    rrWeight = 1.0 / q;
  }
  V3 bounceColor = traceLoop(s, ray0, nextRay, depth - 1, newAttenuation, key, bounce + 1, c, rrStart);   // :74
  V3 clamped = clampV(bounceColor, 0, 10.0);                                 // :75
  if (rrStart > 0) clamped = scale(rrWeight, clamped);
  return add(add(emitted, litColor), clamped);                               // :76
}

}  // namespace

// =================================================================== C API (ctypes)
extern "C" {

struct orc_scene { Scene s; int nNodes = 0, nLeaves = 0, depth = 0; long nPrims = 0; };

// Builds the scene the way parseSceneObjects + constructBVHWithLimit do
// (Rendering/SceneParser.hs:77-78): objects are given in `allObjects` order.
orc_scene *orc_scene_create(int32_t n_objs, const int32_t *kind, const double *data, const int32_t *material,
                            int32_t leaf_threshold, int32_t max_depth,
                            int32_t n_materials, const rt_material *materials,
                            int32_t n_lights, const rt_light *lights,
                            int32_t bg_kind, const double *bg1, const double *bg2,
                            int32_t sky_format, int32_t sky_w, int32_t sky_h, const void *sky_pixels) {
  if (n_objs <= 0) return nullptr;    // BVH.hs:102 error "Cannot compute bounding box of empty list"
  auto *o = new orc_scene();
  std::vector<ObjBox> objs((size_t)n_objs);
  for (int i = 0; i < n_objs; ++i) {
    Prim p; p.kind = kind[i]; std::memcpy(p.d, data + 9 * (size_t)i, sizeof p.d); p.materialId = material[i]; p.id = i;
    objs[i] = {p, boxOf(p)};                                                 // BVH.hs:31
  }
  o->s.bvh = buildBVH(std::move(objs), 0, leaf_threshold, max_depth);        // BVH.hs:32
  for (int i = 0; i < n_materials; ++i) {
    Material m;
    m.diffuse = ld(materials[i].diffuse);
    m.hasEmission = materials[i].flags & RT_MAT_HAS_EMISSION; m.emission = ld(materials[i].emission);
    m.hasShininess = materials[i].flags & RT_MAT_HAS_SHININESS; m.shininess = materials[i].shininess;
    m.hasIor = materials[i].flags & RT_MAT_HAS_IOR; m.ior = materials[i].ior;
    m.hasTransmission = materials[i].flags & RT_MAT_HAS_TRANSMISSION; m.transmission = materials[i].transmission;
    o->s.materials.push_back({materials[i].id, m});
  }
  for (int i = 0; i < n_lights; ++i) o->s.lights.push_back({lights[i].kind, ld(lights[i].vec), ld(lights[i].intensity)});
  o->s.bgKind = bg_kind; o->s.bg1 = ld(bg1); o->s.bg2 = ld(bg2);
  o->s.sky.format = sky_format; o->s.sky.w = sky_w; o->s.sky.h = sky_h;
  size_t n = (size_t)sky_w * sky_h * 3;
  if (sky_format == RT_SKY_RGB8) o->s.sky.u8.assign((const uint8_t *)sky_pixels, (const uint8_t *)sky_pixels + n);
  if (sky_format == RT_SKY_RGBF) o->s.sky.f32.assign((const float *)sky_pixels, (const float *)sky_pixels + n);
  // structure stats
  struct W { static void go(const BVHNode *n, int d, orc_scene *o) {
    o->nNodes++; o->depth = std::max(o->depth, d);
    if (n->leaf) { o->nLeaves++; o->nPrims += (long)n->objs.size(); return; }
    go(n->left.get(), d + 1, o); go(n->right.get(), d + 1, o); } };
  W::go(o->s.bvh.get(), 0, o);
  return o;
}
void orc_scene_destroy(orc_scene *o) { delete o; }
void orc_bvh_info(const orc_scene *o, int32_t *n_nodes, int32_t *n_leaves, int32_t *depth, int64_t *n_prims) {
  *n_nodes = o->nNodes; *n_leaves = o->nLeaves; *depth = o->depth; *n_prims = o->nPrims;
}
// Pre-order flattening in the rt_scene_desc conventions (include/rtcore.h), for comparing the
// product's builder with this tree and for feeding rt_scene_create in tests.
void orc_bvh_export(const orc_scene *o, double *node_bounds, int32_t *node_child,
                    int32_t *prim_kind, double *prim_data, int32_t *prim_material, int32_t *prim_id) {
  int nextNode = 0; int nextPrim = 0;
  struct W { static int go(const BVHNode *n, int &nextNode, int &nextPrim, double *nb, int32_t *nc,
                           int32_t *pk, double *pd, int32_t *pm, int32_t *pi) {
    int me = nextNode++;
    nb[6 * me + 0] = n->box.lo.x; nb[6 * me + 1] = n->box.lo.y; nb[6 * me + 2] = n->box.lo.z;
    nb[6 * me + 3] = n->box.hi.x; nb[6 * me + 4] = n->box.hi.y; nb[6 * me + 5] = n->box.hi.z;
    if (n->leaf) {
      nc[2 * me] = -1 - nextPrim; nc[2 * me + 1] = (int32_t)n->objs.size();
      for (auto &p : n->objs) {
        pk[nextPrim] = p.kind; std::memcpy(pd + 9 * (size_t)nextPrim, p.d, sizeof p.d);
        pm[nextPrim] = p.materialId; pi[nextPrim] = p.id; ++nextPrim;
      }
      return me;
    }
    int l = go(n->left.get(), nextNode, nextPrim, nb, nc, pk, pd, pm, pi);
    int r = go(n->right.get(), nextNode, nextPrim, nb, nc, pk, pd, pm, pi);
    nc[2 * me] = l; nc[2 * me + 1] = r;
    return me;
  } };
  W::go(o->s.bvh.get(), nextNode, nextPrim, node_bounds, node_child, prim_kind, prim_data, prim_material, prim_id);
}

void orc_default_camera(const double *lookFrom, const double *lookAt, const double *vUp, double vfov, double aspect,
                        double aperture, double focusDist, rt_camera *out) {
  *out = defaultCamera(ld(lookFrom), ld(lookAt), ld(vUp), vfov, aspect, aperture, focusDist);
}

struct orc_counters {
  uint64_t rays_closest, rays_shadow, nodes_literal, nodes_nonempty;
  uint64_t tri_tests, sphere_tests, plane_tests;
  uint64_t tri_tests_nonempty, sphere_tests_nonempty, plane_tests_nonempty;
  uint64_t shaded_hits;
};

// The render driver: worker/renderRow/pixelColor/samplePixel, Rendering/RenderWorker.hs:53-114,
// rows row0, row0+row_stride, ... < row1 of the FILE (row j of the file is rowIdx = H-1-j, :66) over `n_threads` threads
// pulling rows from a shared counter (the TQueue of ImageGenerator.hs:64-71).
// Outputs are full-frame sized (3*W*H etc.); only the requested rows are written.
int orc_render(const orc_scene *o, const rt_camera *cam, const rt_params *prm,
               int32_t row0, int32_t row1, int32_t row_stride, int32_t n_threads,
               double *radiance, double *color, uint8_t *rgb8, int32_t *prim_id, orc_counters *counters) {
  const Scene &s = o->s;
  const int W = prm->width, H = prm->height, spp = prm->samples_per_pixel;
  if (W <= 0 || H <= 0 || spp <= 0 || row0 < 0 || row1 > H || row_stride < 1) return RT_ERR_INVALID;
  std::atomic<int> nextRow(0);
  std::vector<Counters> perThread((size_t)std::max(1, n_threads));
  auto work = [&](int tid) {
    Counters *c = counters ? &perThread[tid] : nullptr;
    for (;;) {
      int j = row0 + row_stride * nextRow.fetch_add(1);                    // tryReadTQueue, RenderWorker.hs:55
      if (j >= row1) break;
      int rowIdx = H - 1 - j;                                              // :66
      for (int i = 0; i < W; ++i) {                                        // :67
        RngKey key; key.key[0] = (uint32_t)prm->seed; key.key[1] = (uint32_t)(prm->seed >> 32);
        key.pixel = (uint32_t)((size_t)j * W + i);
        V3 colsum{0, 0, 0};                                                // averageColor foldl' add 0, :111-114
        for (int sIdx = 0; sIdx < spp; ++sIdx) {                           // replicateM spp, :78
          key.sample = (uint32_t)sIdx;
          Ray ray = generateRay(*cam, i, rowIdx, W, H, prm->antialiasing != 0, key);   // :102
          if (sIdx == 0 && prim_id) {   // id of the first traced ray's hit; nothing is traced when maxBounces = 0
            HitRecord r0;
            bool h = prm->max_bounces > 0 && hitBVH(s.bvh.get(), ray, {0.001, 10000}, r0, nullptr);
            prim_id[(size_t)j * W + i] = h ? r0.primId : -1;
          }
          V3 col = traceLoop(s, ray, ray, prm->max_bounces, V3{1, 1, 1}, key, 0, c, prm->rr_start_bounce);   // :106, Trace.hs:20
          colsum = add(colsum, col);
        }
        V3 averaged = scale(1.0 / (double)spp, colsum);                    // :114
        V3 toned = toneMap(prm->exposure, averaged);                       // :80
        V3 gammaed = gammaCorrect(prm->gamma, toned);                      // :81
        size_t k = ((size_t)j * W + i) * 3;
        if (radiance) { radiance[k] = averaged.x; radiance[k + 1] = averaged.y; radiance[k + 2] = averaged.z; }
        if (color) { color[k] = gammaed.x; color[k + 1] = gammaed.y; color[k + 2] = gammaed.z; }
        if (rgb8) {                                                        // showPixel, :108-109 truncate (c*255.999)
          double ch[3] = {gammaed.x, gammaed.y, gammaed.z};
          for (int q = 0; q < 3; ++q) {
            double v = std::trunc(ch[q] * 255.999);
            rgb8[k + q] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : (v != v ? 0 : v));
          }
        }
      }
    }
  };
  if (n_threads <= 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
    for (auto &t : th) t.join();
  }
  if (counters) {
    Counters tot; for (auto &c : perThread) tot += c;
    *counters = {tot.rays_closest, tot.rays_shadow, tot.nodes_literal, tot.nodes_nonempty,
                 tot.tri_tests, tot.sphere_tests, tot.plane_tests,
                 tot.tri_tests_nonempty, tot.sphere_tests_nonempty, tot.plane_tests_nonempty, tot.shaded_hits};
  }
  return RT_OK;
}

// ----------------------------------------------------------------- unit-level entry points for the KATs
int orc_hit_aabb(const double *lo, const double *hi, const double *o, const double *d, double tmin, double tmax, double *out2) {
  Interval r; bool h = hitAABBInterval({ld(lo), ld(hi)}, {ld(o), ld(d)}, {tmin, tmax}, r);
  if (h) { out2[0] = r.lo; out2[1] = r.hi; }
  return h;
}
// out = t, point[3], normal[3], frontFace
int orc_hit_prim(int32_t kind, const double *data9, const double *o, const double *d, double tmin, double tmax, double *out8) {
  Prim p; p.kind = kind; std::memcpy(p.d, data9, sizeof p.d); p.materialId = 0; p.id = 0;
  HitRecord r; bool h = hitPrim(p, {ld(o), ld(d)}, {tmin, tmax}, r, nullptr);
  if (h) { out8[0] = r.t; out8[1] = r.point.x; out8[2] = r.point.y; out8[3] = r.point.z;
           out8[4] = r.normal.x; out8[5] = r.normal.y; out8[6] = r.normal.z; out8[7] = r.frontFace; }
  return h;
}
void orc_prim_box(int32_t kind, const double *data9, double *out6) {
  Prim p; p.kind = kind; std::memcpy(p.d, data9, sizeof p.d);
  AABB b = boxOf(p);
  out6[0] = b.lo.x; out6[1] = b.lo.y; out6[2] = b.lo.z; out6[3] = b.hi.x; out6[4] = b.hi.y; out6[5] = b.hi.z;
}
// closest hit through the BVH: returns prim id or -1; out = t, point, normal, frontFace
int orc_hit_scene(const orc_scene *o, const double *org, const double *dir, double tmin, double tmax, double *out8) {
  HitRecord r; bool h = hitBVH(o->s.bvh.get(), {ld(org), ld(dir)}, {tmin, tmax}, r, nullptr);
  if (!h) return -1;
  out8[0] = r.t; out8[1] = r.point.x; out8[2] = r.point.y; out8[3] = r.point.z;
  out8[4] = r.normal.x; out8[5] = r.normal.y; out8[6] = r.normal.z; out8[7] = r.frontFace;
  return r.primId;
}
double orc_schlick(double c, double r) { return schlick(c, r); }
void orc_reflect(const double *v, const double *n, double *out) { V3 r = reflect(ld(v), ld(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_refract(const double *v, const double *n, double eta, double *out) { V3 r = refract(ld(v), ld(n), eta); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
void orc_tonemap_gamma(double exposure, double gamma, const double *c, double *out) {
  V3 r = gammaCorrect(gamma, toneMap(exposure, ld(c))); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_background(const orc_scene *o, const double *dir, double *out) {
  V3 r = backgroundSample(o->s, V3{1, 1, 1}, Ray{V3{0, 0, 0}, ld(dir)}); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_generate_ray(const rt_camera *cam, int32_t i, int32_t j, int32_t W, int32_t H, int32_t aa, uint64_t seed,
                      uint32_t pixel, uint32_t sample, double *out6) {
  RngKey key; key.key[0] = (uint32_t)seed; key.key[1] = (uint32_t)(seed >> 32); key.pixel = pixel; key.sample = sample;
  Ray r = generateRay(*cam, i, j, W, H, aa != 0, key);
  out6[0] = r.o.x; out6[1] = r.o.y; out6[2] = r.o.z; out6[3] = r.d.x; out6[4] = r.d.y; out6[5] = r.d.z;
}
void orc_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10(ctr, key, out); }
// one path, for statistical tests of traceRay
void orc_trace_path(const orc_scene *o, const double *org, const double *dir, int32_t max_bounces, uint64_t seed,
                    uint32_t pixel, uint32_t sample, double *out3) {
  RngKey key; key.key[0] = (uint32_t)seed; key.key[1] = (uint32_t)(seed >> 32); key.pixel = pixel; key.sample = sample;
  Ray r{ld(org), ld(dir)};
  V3 c = traceLoop(o->s, r, r, max_bounces, V3{1, 1, 1}, key, 0, nullptr);
  out3[0] = c.x; out3[1] = c.y; out3[2] = c.z;
}

}  // extern "C"
