// rt_device.h — device data layout and FP64 vector math of the MI355X render core.
//
// Layout in HBM (DESIGN.md §Data layout):
//   DNode<float>  64 B : both children's boxes as conservatively rounded FP32 + 2 refs
//   DNode<double> 128 B: both children's boxes as the reference's FP64 (exact slab test; triage / parity mode)
//   DPrim         80 B : 9 doubles + material index + kind/last-in-leaf flag, stored in leaf order
//   DMaterial     64 B, DLight 64 B
// A ref >= 0 is an internal node index; a ref < 0 is a leaf whose first primitive is ~ref and
// whose last primitive carries DPRIM_LAST.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DPRIM_KIND_MASK 0x3u
#define DPRIM_LAST 0x80000000u

#define DMAT_GLASS_DIR 1u   // transmission == Just 1.0 && ior present      (Trace.hs:48-49)
#define DMAT_NO_ATTEN  2u   // transmission == Just 1.0                      (Trace.hs:69-71)
#define DMAT_METAL     4u   // shininess present && > 100                    (Trace.hs:60-61)

struct alignas(16) DPrim {
  double d[9];     // triangle: p0, e1 = p1-p0, e2 = p2-p0 ; sphere: c, r ; plane: p, n, leaf-box index
  int32_t mat;     // dense material index (0 = defaultMaterial)
  uint32_t kind;   // RT_PRIM_* | DPRIM_LAST
};
static_assert(sizeof(DPrim) == 80, "DPrim must be 80 bytes");

template <typename BT> struct DNode;
template <> struct alignas(64) DNode<float> {
  float box[12];   // [child][lo.xyz, hi.xyz]
  int32_t ref[2];
  int32_t pad[2];
};
template <> struct alignas(128) DNode<double> {
  double box[12];
  int32_t ref[2];
  int32_t pad[6];
};
static_assert(sizeof(DNode<float>) == 64, "FP32 node must be 64 bytes");
static_assert(sizeof(DNode<double>) == 128, "FP64 node must be 128 bytes");

struct alignas(16) DMaterial {
  double diffuse[3];
  double emission[3];
  double ior;
  uint32_t flags;
  uint32_t pad;
};
static_assert(sizeof(DMaterial) == 64, "DMaterial must be 64 bytes");

struct alignas(16) DLight {
  double vec[3];
  double intensity[3];
  int32_t kind;
  int32_t pad[3];
};
static_assert(sizeof(DLight) == 64, "DLight must be 64 bytes");

// ---------------------------------------------------------------- FP64 3-vectors (Core/Vec3.hs:53-111)
struct d3 { double x, y, z; };
#define RT_DEV __device__ __forceinline__
#ifndef RT_NOINLINE
#define RT_NOINLINE __attribute__((noinline))   // out-of-line helpers (rare paths); -DRT_NOINLINE= inlines them (experiment)
#endif
RT_DEV d3 mk3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_DEV d3 ld3(const double *p) { return mk3(p[0], p[1], p[2]); }
RT_DEV d3 operator+(d3 a, d3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_DEV d3 operator-(d3 a, d3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_DEV d3 operator*(d3 a, d3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_DEV d3 operator*(double s, d3 v) { return mk3(s * v.x, s * v.y, s * v.z); }
RT_DEV d3 neg3(d3 v) { return mk3(-v.x, -v.y, -v.z); }
RT_DEV double dot3(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_DEV d3 cross3(d3 a, d3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_DEV d3 normalize3(d3 v) { return (1.0 / sqrt(dot3(v, v))) * v; }                 // Vec3.hs:78-79
RT_DEV d3 reflect3(d3 v, d3 n) { return v - (2.0 * dot3(v, n)) * n; }               // Vec3.hs:98-99
RT_DEV d3 refract3(d3 v, d3 n, double eta) {                                         // Vec3.hs:102-107
  double cosTheta = -dot3(v, n);
  d3 perp = eta * (v + cosTheta * n);
  d3 par = (-sqrt(fabs(1.0 - dot3(perp, perp)))) * n;
  return perp + par;
}
// clamp, Trace.hs:91-94: `max lo (min hi v)` with GHC's Ord Double defaults - max x y = if x <= y then y else x, min x y = if x <= y
// then x else y - which is NOT fmax / fmin on a NaN: min hi NaN = NaN, max lo NaN = lo, so the reference scrubs a NaN bounce term
// (0 x an infinite sky texel) to 0 where fmax(lo, fmin(hi, NaN)) would return hi = 10 (VERDICT r02 "missing" #6).
RT_DEV double hs_min(double x, double y) { return x <= y ? x : y; }
RT_DEV double hs_max(double x, double y) { return x <= y ? y : x; }
RT_DEV double clampd(double v, double lo, double hi) { return hs_max(lo, hs_min(hi, v)); }
RT_DEV d3 clamp3(d3 v, d3 lo, d3 hi) { return mk3(clampd(v.x, lo.x, hi.x), clampd(v.y, lo.y, hi.y), clampd(v.z, lo.z, hi.z)); }
// The same on values that are never NaN - the offsets and bounds of the pending clamp function, sums of emission and light terms -
// where v_max / v_min give the same numbers in a third of the instructions (the GHC form on every fold cost 0.3 ms of C4's 41).
// A NaN can only enter a path through its LAST term (0 x an infinite sky texel), which finish_paths clamps with clamp3.
RT_DEV d3 clamp3_finite(d3 v, d3 lo, d3 hi) {
  return mk3(fmax(lo.x, fmin(hi.x, v.x)), fmax(lo.y, fmin(hi.y, v.y)), fmax(lo.z, fmin(hi.z, v.z)));
}

// ---------------------------------------------------------------- Philox4x32-10
struct u4 { uint32_t x, y, z, w; };
RT_DEV u4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
  return o;
}
RT_DEV double u01(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
RT_DEV double um11(uint32_t w) { return ((double)w + 0.5) * (1.0 / 2147483648.0) - 1.0; }
