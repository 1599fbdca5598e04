// rt_kernels.h — device code of the MI355X render core: shared structs, the reference's per-ray functions
// (camera, background, box and primitive tests), the render kernel (rt_render_kernel.h) and the small
// resolve / un-tile / quantise kernels.
//
// Reference semantics restated here (paths relative to /root/reference/app):
//   camera_ray            Rendering/Camera.hs:47-60, Rendering/RenderWorker.hs:95-106
//   traversal             Hittable/BVH.hs:120-143, Hittable/BoundingBox.hs:18-37
//   primitive tests       Hittable/Objects/{Triangle.hs:30-64, Sphere.hs:24-40, Plane.hs:21-32}
//   shadow rays           Rendering/Light.hs:16-49
//   material bounce       Rendering/Trace.hs:22-76
//   background / sky      Rendering/ImageGenerator.hs:130-135, Rendering/SkySphere.hs:32-47
//   tone map, gamma       Rendering/Trace.hs:96-105
#pragma once
#include "rt_device.h"
#include "../../include/rtcore.h"
#include "rt_sky_tables.h"
#include "rt_pow5.h"

struct KScene {
  const void *nodes;           // DNode<BT>[] of the tree the kernel traverses
  const DPrim *prims;          // primitives in the reference's leaf order (shading reads these)
  const DPrim *tprims;         // primitives in the traversed tree's leaf order (FAST: .mat = index into prims; else == prims)
  // FAST only: the reference tree, consulted when two primitives are hit at exactly the same distance
  const void *ref_nodes64;     // DNode<double>[]: both children's FP64 boxes per internal node
  const int32_t *ref_anc;      // per internal node: (parent << 1) | side (0 = left child), root: -1
  const int32_t *ref_depth;    // per internal node: edges from the root
  const int32_t *leaf_anc;     // per primitive: (parent of its leaf << 1) | side
  const int32_t *prim_ids;
  const DMaterial *mats;
  const DLight *lights;
  const double *plane_boxes;   // 6 doubles per plane: the FP64 box of the leaf holding it
  // What shading reads of a hit, ONE 64-byte line per primitive (reference order): doubles 0-2 n = normalize(e1 x e2) (Triangle.hs:51),
  // 3-5 normalize(n) (Light.hs:19) (triangles only), then the int pair (material index, kind) of DPrim; 8 bytes of padding.
  // (The pair used to be read from the 80-byte primitive and the normals from a 48-byte table: three lines in two arrays per hit.)
  const double *shade_recs;
  const void *sky;
  const double *sky_u_tab;     // (cos phi_k, sin phi_k), phi_k = 2 pi (k / w - 0.5), k = -2 .. w + 2: the texel columns' boundary directions
  const double *sky_v_tab;     // T_k, k = -2 .. h + 2 (+inf up to 0, -inf above h): the largest y the reference puts in a row >= k (rt_sky_tables.h)
  const double *sky_u_q;       // Q_k[3]: per column boundary and region of (x, z), the smallest z / x the reference puts in a column >= k (rt_sky_tables.h)
  int32_t root_ref;
  int32_t ref_root_ref;        // root of the reference's tree (== root_ref unless FAST)
  int32_t n_lights;
  int32_t bg_kind, sky_format, sky_w, sky_h;
  double bg1[3], bg2[3];
  float rscene;                // max |coordinate| of the root box
};

struct KFrame {
  rt_camera cam;
  int32_t W, H, spp, aa, maxb;
  double exposure, gamma;
  uint32_t key0, key1;
  const int32_t *tiles;        // this rank's tile ids, or NULL = all tiles in order
  int32_t n_tiles, tiles_x;
  int32_t packed;              // 1: outputs indexed by local_tile*64+position ; 0: row-major full frame
  double *out_color, *out_radiance;
  uint8_t *out_rgb8;           // the 8-bit frame (showPixel, RenderWorker.hs:108-109), written by k_resolve in the same pass
  int32_t *out_primid;
  uint32_t *tile_counter;      // hands out chunks of items
  unsigned long long *stats;   // STAT_COUNT counters, see STAT_*
  double *fscratch;            // [grid waves][scratch_rows][64] x 8 B: the records' cold rows (PR_ROWS), then the stack overflow rows
  int32_t scratch_rows;        // PR_ROWS + max(0, stack_depth - RT_STACK_LDS)
  int32_t stack_depth;
  int32_t top_nodes;           // the first top_nodes internal nodes (top levels, breadth-first) are cached in LDS
  // one launch covers the local tiles [tile_begin, tile_end) of this rank's list: items = (tile, position, sample)
  int32_t tile_begin, tile_end;
  uint32_t n_items;            // (tile_end - tile_begin) * 64 * spp
  uint32_t chunk;              // items a wave takes from the counter at a time (a multiple of 64, set per launch)
  double *samples;             // [n_items][3] radiance of every path of this launch, summed per pixel by k_resolve
  int32_t rr_start;            // 0: off (the reference); k > 0: Russian roulette once a path has made k bounces (opt-in extension)
  int32_t lit_row;             // first row of literal_hit's per-lane stack within a wave's fscratch block (scenes with planes), else -1
  int32_t exact_row;           // BT = double: first of stack_depth rows holding the stacked children's exact FP64 entry distances
  unsigned long long *dbg;     // experiment builds (-DRT_TIMERS) only: 4 words per wave, see the end of k_render_tiles
};

enum { STAT_CLOSEST = 0, STAT_SHADOW, STAT_NODES, STAT_TRI, STAT_SPHERE, STAT_PLANE, STAT_SHADED, STAT_IT_NODE, STAT_IT_PRIM, STAT_IT_SHADE, STAT_LIVE, STAT_TOPHIT, STAT_SHADOW_CULLED, STAT_LITERAL,
       STAT_SH_L0, STAT_SH_L1, STAT_SH_L2, STAT_SH_L3, STAT_SH_L4,   // lanes that ran SHADE's sub-blocks: hit, light term, shadow set-up, scatter, finish
       STAT_SH_W0, STAT_SH_W1, STAT_SH_W2, STAT_SH_W3, STAT_SH_W4,   // SHADE trips in which at least one lane ran them
       STAT_T_VOTE, STAT_T_NODE, STAT_T_PRIM, STAT_T_SHADE, STAT_T_REFILL,
       STAT_T_SH0, STAT_T_SH1, STAT_T_SH2, STAT_T_SH3, STAT_T_SH4, STAT_T_SH5, STAT_T_SH6,   // -DRT_TIMERS: SHADE's time by sub-block (loads, hit, light, set-up, scatter, finish, begin query)
       STAT_COUNT };

// cold code (once per path, not per traversal step).  Inlined: taking KScene/KFrame by reference out of line cost more
// scratch than it saved registers; only the libm-heavy leaves (sky_texel_exact, pow_gamma) are real calls.
#define RT_COLD __device__ __forceinline__

#define RT_TMIN 0.001
#define RT_TMAX 10000.0

// ---------------------------------------------------------------- camera (Camera.hs:47-60)
RT_COLD void camera_ray(const KFrame &fr, int px, int py, uint32_t pixel, uint32_t sample, d3 &o, d3 &d) {
  double uo = 0.5, vo = 0.5, lx = 0.0, ly = 0.0;                  // RenderWorker.hs:98-99
  const double ap = fr.cam.aperture;
  if (fr.aa || ap != 0.0) {
    u4 w = philox4x32_10(pixel, sample, 0u, 0u, fr.key0, fr.key1);
    if (fr.aa) { uo = u01(w.x); vo = u01(w.y); }
    if (ap != 0.0) {                                              // randomInUnitDisk, Vec3.hs:126-133
      lx = um11(w.z); ly = um11(w.w);
      uint32_t sub = 1;
      while (!(lx * lx + ly * ly < 1.0) && sub < 64u) {
        w = philox4x32_10(pixel, sample, 0u, sub, fr.key0, fr.key1);
        lx = um11(w.x); ly = um11(w.y);
        if (!(lx * lx + ly * ly < 1.0)) { lx = um11(w.z); ly = um11(w.w); }
        ++sub;
      }
      if (!(lx * lx + ly * ly < 1.0)) { lx = 0.0; ly = 0.0; }
    }
  }
  d3 lens = (ap / 2) * (lx * ld3(fr.cam.u) + ly * ld3(fr.cam.v));   // :50
  o = ld3(fr.cam.origin) + lens;                                    // :51
  double rayu = ((double)px + uo) / (double)fr.W;                   // :54
  double rayv = ((double)(fr.H - 1 - py) + vo) / (double)fr.H;      // :55 with j = rowIdx (RenderWorker.hs:66)
  d3 dir = ((ld3(fr.cam.lower_left) + rayu * ld3(fr.cam.horizontal)) + rayv * ld3(fr.cam.vertical)) - o;   // :56-59
  d = normalize3(dir);                                              // :60
}

#define SKY_NEAR_BOUNDARY 1.6e-14   // |cross product| below which a column is decided by its quotient z / x; the pipeline's rounding is worth ~3e-15
#define SKY_TAB_PAD 2      // the tables start at k = -SKY_TAB_PAD

// GHC's atan2 on Double: the RealFloat class default of GHC.Float (the Haskell 2010 report's definition; instance RealFloat
// Double does not override it), NOT libm's atan2 - it differs from it in the last bits (division, atan and pi + ... each round).
__device__ inline double hs_atan2(double y, double x) {
  const double pi = 3.14159265358979323846;
  if (x > 0.0) return atan(y / x);
  if (x == 0.0 && y > 0.0) return pi / 2;
  if (x < 0.0 && y > 0.0) return pi + atan(y / x);
  const bool nzy = y == 0.0 && signbit(y), nzx = x == 0.0 && signbit(x);
  if ((x <= 0.0 && y < 0.0) || (x < 0.0 && nzy) || (nzx && nzy)) {   // = -atan2 (-y) x, whose guards are the three above and `pi`
    const double ny = -y;
    return -(x == 0.0 && ny > 0.0 ? pi / 2 : (x < 0.0 && ny > 0.0 ? pi + atan(ny / x) : pi));
  }
  if (y == 0.0 && (x < 0.0 || nzx)) return pi;
  if (x == 0.0 && y == 0.0) return y;
  return x + y;
}

// The reference's texel for ANY direction, the slow way - for the lanes sky_texel's table tests cannot settle (a direction within
// rounding of a column boundary, a zero x or z, a failed estimate).  ONE out-of-line call with few arguments, so that the FP64
// atan / asin bodies (~100 temporaries) and the FP64 division do not set the register allocation of the render kernel.
// Column: the reference's own expression (SkySphere.hs:35,39) with ocml's atan is right to within one column; which of the
// three it is, is decided the way the reference's roundings decide it, by comparing z / x with the thresholds of boundaries i and
// i + 1 (skytab::side_q, rt_sky_tables.h).  With a zero x or z (atan2's other guards; they give exact multiples of pi / 2) or at
// the +-pi cut (no table entry; the column saturates at 0 / w - 1) the expression's value stands.  Row: likewise, against T_k.
__device__ RT_NOINLINE int2 sky_texel_exact(const double *u_q, double x, double y, double z) {
  const int w = (int)u_q[-2], h = (int)u_q[-1];                           // one table: (w, h), Q_k[3] ..., T_k ...  (rtcore.hip)
  const double *v_tab = u_q + 3 * (w + 2 * SKY_TAB_PAD + 1);
  const double dy = fmax(-1.0, fmin(1.0, y));
  const double u = 0.5 + hs_atan2(z, x) / (2 * 3.14159265358979323846);   // :35
  const double v = 0.5 - asin(dy) / 3.14159265358979323846;               // :36
  long long i = (long long)floor(u * (double)w), j = (long long)floor(v * (double)h);
  i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
  j = j < 0 ? 0 : (j > h - 1 ? h - 1 : j);
  {
    const double *Q = u_q + 3 * ((int)i + SKY_TAB_PAD);
    const int s0 = i >= 1 ? skytab::side_q(Q, x, z) : 1, s1 = i + 1 <= w - 1 ? skytab::side_q(Q + 3, x, z) : 0;
    if (s0 >= 0 && s1 >= 0) i = i - 1 + s0 + s1;
  }
  if (dy == dy) {
    const double *T = v_tab + ((int)j + SKY_TAB_PAD);
    j = j - 1 + (dy <= T[0] ? 1 : 0) + (dy <= T[1] ? 1 : 0);
  }
  i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
  j = j < 0 ? 0 : (j > h - 1 ? h - 1 : j);
  return make_int2((int)i, (int)j);
}

// Texel of a unit direction (SkySphere.hs:35-41): i = clamp(floor(u w)), j = clamp(floor(v h)) with u = 0.5 + atan2 z x / 2pi,
// v = 0.5 - asin(y) / pi.  Only the INTEGERS matter, so the two FP64 libm calls (~400 VALU instructions, once per path) are
// replaced by their inverse, tabulated on the host at upload (rt_sky_tables.h):
//   rows     floor(v h) >= k  <=>  y <= T_k, T_k the largest DOUBLE for which the reference's expression (with the host's asin)
//            gives a row >= k: exact for every y, boundary or not;
//   columns  floor(u w) >= k  <=>  the direction (x, z) lies counter-clockwise of the boundary direction (cos phi_k, sin phi_k),
//            phi_k = 2pi (k / w - 0.5), i.e. cos phi_k z - sin phi_k x >= 0 (valid while the two are less than pi apart) - unless the
//            direction is within rounding of the boundary, where the reference's texel is decided by how the division, atan, the
//            sums and the product of GHC's atan2 and of u w round: there sky_texel_exact compares z / x with the boundary's threshold Q_k.
// A cheap FP32 estimate gives the texel to within one; four threshold tests around it give the texel.  If the estimate is ever off
// by more than one texel (or NaN), or the direction has a zero x or z or is at the +-pi cut, the lane evaluates the reference's
// expressions themselves (sky_texel_exact).
RT_COLD void sky_texel(const double *u_tab, const double *u_q, const double *v_tab, int w, int h, d3 n, int &i, int &j) {
  const double dy = fmax(-1.0, fmin(1.0, n.y));
  bool ok = true;
  {                                                                 // ---- column: FP32 estimate of atan2(z, x)
    const float x = (float)n.x, z = (float)n.z, ax = fabsf(x), az = fabsf(z);
    const float mx = fmaxf(ax, az), mn = fminf(ax, az);
    const float a = mn * __frcp_rn(mx), q = a * a;
    float r = a * (0.99997726f + q * (-0.33262347f + q * (0.19354346f + q * (-0.11643287f + q * (0.05265332f + q * -0.01172120f)))));
    r = az > ax ? 1.57079637f - r : r;
    r = x < 0.0f ? 3.14159274f - r : r;
    r = z < 0.0f ? -r : r;
    int i0 = (int)floorf((0.5f + r * 0.159154943f) * (float)w);
    i0 = i0 < 0 ? 0 : (i0 > w - 1 ? w - 1 : i0);
    const double2 *t = (const double2 *)u_tab + (i0 - 1 + SKY_TAB_PAD);                 // boundaries i0-1 .. i0+2
    const double2 b0 = t[0], b1 = t[1], b2 = t[2], b3 = t[3];
    const double v1 = b1.x * n.z - b1.y * n.x, v2 = b2.x * n.z - b2.y * n.x;
    const bool c0 = b0.x * n.z - b0.y * n.x >= 0.0, c1 = v1 >= 0.0;
    const bool c2 = v2 >= 0.0, c3 = b3.x * n.z - b3.y * n.x >= 0.0;
    // zeros: atan2's other guards (+pi or -pi by the sign of a zero ...), which a cross product cannot see
    ok = ok && c0 && !c3 && n.z != 0.0 && n.x != 0.0 && (mx == mx) && mx > 0.0f;
    // A direction within rounding of a boundary belongs to whichever texel the reference's FLOATING-POINT u w falls into, not to the
    // one exact geometry says.  This is not a measure-zero corner: a camera that looks exactly along an axis has a centre column of
    // rays with x or z = +-1e-16 (rounding residue), and x = +-z on its diagonals (tools/fuzz_oracle.py --extreme,
    // tools/sky_boundary_probe.py).  Such a lane asks sky_texel_exact.
    ok = ok && fabs(v1) > SKY_NEAR_BOUNDARY && fabs(v2) > SKY_NEAR_BOUNDARY;
    i = i0 - 1 + (c1 ? 1 : 0) + (c2 ? 1 : 0);
  }
  {                                                                 // ---- row: FP32 estimate of asin(y) (Abramowitz-Stegun 4.4.45)
    const float y = (float)dy, ay = fabsf(y);
    float p = sqrtf(fmaxf(0.0f, 1.0f - ay)) * (1.5707288f + ay * (-0.2121144f + ay * (0.0742610f + ay * -0.0187293f)));
    p = 1.57079637f - p;
    p = y < 0.0f ? -p : p;
    int j0 = (int)floorf((0.5f - p * 0.318309886f) * (float)h);
    j0 = j0 < 0 ? 0 : (j0 > h - 1 ? h - 1 : j0);
    const double *t = v_tab + (j0 - 1 + SKY_TAB_PAD);                                   // thresholds j0-1 .. j0+2
    const bool c0 = dy <= t[0], c1 = dy <= t[1], c2 = dy <= t[2], c3 = dy <= t[3];
    ok = ok && c0 && (!c3 || j0 + 2 > h);                           // (dy = -1 reaches the last threshold: clamped below)
    j = j0 - 1 + (c1 ? 1 : 0) + (c2 ? 1 : 0);
  }
  if (!ok) {                                                        // rare: see sky_texel_exact
    const int2 r = sky_texel_exact(u_q, n.x, n.y, n.z);
    i = r.x; j = r.y;
  }
  i = i < 0 ? 0 : (i > w - 1 ? w - 1 : i);
  j = j < 0 ? 0 : (j > h - 1 ? h - 1 : j);
}

#ifdef RT_SKY_OOL   // experiment: the whole lookup out of line
__device__ RT_NOINLINE int2 sky_texel_ool(const double *u_tab, const double *u_q, const double *v_tab, int w, int h, double x, double y, double z) {
  int i, j;
  sky_texel(u_tab, u_q, v_tab, w, h, mk3(x, y, z), i, j);
  return make_int2(i, j);
}
#endif

// ---------------------------------------------------------------- background (Trace.hs:78-83)
RT_COLD d3 background(const KScene &sc, d3 dir) {
  if (sc.sky_format != RT_SKY_NONE) {                               // SkySphere.hs:32-47
    d3 n = normalize3(dir);
    int i, j;
#ifdef RT_SKY_OOL
    { const int2 r = sky_texel_ool(sc.sky_u_tab, sc.sky_u_q, sc.sky_v_tab, sc.sky_w, sc.sky_h, n.x, n.y, n.z); i = r.x; j = r.y; }
#else
    sky_texel(sc.sky_u_tab, sc.sky_u_q, sc.sky_v_tab, sc.sky_w, sc.sky_h, n, i, j);
#endif
    size_t k = ((size_t)j * sc.sky_w + (size_t)i) * 3;
    if (sc.sky_format == RT_SKY_RGB8) {
      const uint8_t *p = (const uint8_t *)sc.sky;
      return mk3(p[k] / 255.0, p[k + 1] / 255.0, p[k + 2] / 255.0);
    }
    const float *p = (const float *)sc.sky;
#ifdef RT_SKY_NT
    return mk3((double)__builtin_nontemporal_load(p + k), (double)__builtin_nontemporal_load(p + k + 1), (double)__builtin_nontemporal_load(p + k + 2));
#else
    return mk3((double)p[k], (double)p[k + 1], (double)p[k + 2]);
#endif
  }
  if (sc.bg_kind == RT_BG_SOLID) return ld3(sc.bg1);                // ImageGenerator.hs:131
  d3 unit = normalize3(dir);                                        // :133
  double t = 0.5 * (unit.y + 1.0);                                  // :134
  return (1.0 - t) * ld3(sc.bg2) + t * ld3(sc.bg1);                 // :135, Color.hs:16-17
}

// ---------------------------------------------------------------- ray/box set-up
// BoxRay<double>: the reference's slab test verbatim (BoundingBox.hs:18-37).
// BoxRay<float> : a conservative FP32 superset of it (DESIGN.md §Conservative FP32 boxes): boxes are
//   rounded outward at upload, the ray origin is widened by a margin m that covers the FP32 rounding of
//   origin and direction, and the enter/exit comparison carries a relative slack, so a box the FP64
//   test accepts is never rejected here.  Primitive tests stay FP64 either way.
template <typename BT> struct BoxRay;
template <> struct BoxRay<double> {
  double ox, oy, oz, ix, iy, iz;
  bool px, py, pz;
};
template <> struct BoxRay<float> {
  float ix, iy, iz, ax, ay, az, bx, by, bz;
};

RT_DEV void make_boxray(BoxRay<double> &b, d3 o, d3 d, float) {
  b.ox = o.x; b.oy = o.y; b.oz = o.z;
  b.px = fabs(d.x) < 1e-8; b.py = fabs(d.y) < 1e-8; b.pz = fabs(d.z) < 1e-8;      // :21
  b.ix = 1.0 / d.x; b.iy = 1.0 / d.y; b.iz = 1.0 / d.z;                           // :24
}
#if defined(__HIP_DEVICE_COMPILE__)
RT_DEV float rcp_approx(float v) { return __builtin_amdgcn_rcpf(v); }
#else
RT_DEV float rcp_approx(float v) { return 1.0f / v; }       // (the host pass only parses the kernels)
#endif
RT_DEV float safe_dir(float v) { return fabsf(v) < 1e-30f ? (v < 0.0f ? -1e-30f : 1e-30f) : v; }
RT_DEV void make_boxray(BoxRay<float> &b, d3 o, d3 d, float rscene) {
  float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z;
  float omax = fmaxf(fabsf(ox), fmaxf(fabsf(oy), fabsf(oz)));
  // reachable region: |coord| <= omax + 1e4 (t <= 10000, |d| ~ 1), clipped to the scene
  float m = 9.5367431640625e-7f * (fminf(rscene, omax + 1.0e4f) + omax) + 1e-30f;   // 2^-20 * extent
  // 1 / d by v_rcp_f32 (1 ulp) instead of an IEEE division (10 instructions each): the factor (1 + e) multiplies a whole slab
  // distance (lo - o -+ m) * i, and e * |lo - o| <= 2^-23 * extent is an eighth of the margin m
  b.ix = rcp_approx(safe_dir((float)d.x)); b.iy = rcp_approx(safe_dir((float)d.y)); b.iz = rcp_approx(safe_dir((float)d.z));
  b.ax = (ox + m) * b.ix; b.ay = (oy + m) * b.iy; b.az = (oz + m) * b.iz;
  b.bx = (ox - m) * b.ix; b.by = (oy - m) * b.iy; b.bz = (oz - m) * b.iz;
}

// Slab test of one child box against [tmin, tmax]; returns hit and the entry distance.
RT_DEV bool box_test(const BoxRay<double> &r, const double *bx, double tmin, double tmax, double &tenter) {
  double a0, a1, b0, b1, c0, c1;
  const double inf = __builtin_huge_val();
  bool ok = true;
  if (r.px) { ok = ok && !(r.ox < bx[0] || r.ox > bx[3]); a0 = -inf; a1 = inf; }
  else { double t0 = (bx[0] - r.ox) * r.ix, t1 = (bx[3] - r.ox) * r.ix; a0 = fmin(t0, t1); a1 = fmax(t0, t1); }
  if (r.py) { ok = ok && !(r.oy < bx[1] || r.oy > bx[4]); b0 = -inf; b1 = inf; }
  else { double t0 = (bx[1] - r.oy) * r.iy, t1 = (bx[4] - r.oy) * r.iy; b0 = fmin(t0, t1); b1 = fmax(t0, t1); }
  if (r.pz) { ok = ok && !(r.oz < bx[2] || r.oz > bx[5]); c0 = -inf; c1 = inf; }
  else { double t0 = (bx[2] - r.oz) * r.iz, t1 = (bx[5] - r.oz) * r.iz; c0 = fmin(t0, t1); c1 = fmax(t0, t1); }
  tenter = fmax(fmax(a0, b0), fmax(c0, tmin));                     // :30
  double texit = fmin(fmin(a1, b1), fmin(c1, tmax));               // :31
  return ok && tenter <= texit;                                    // :32
}
#ifndef RT_SLACK
#define RT_SLACK 1.000001f
#endif
RT_DEV bool box_test(const BoxRay<float> &r, const float *bx, float tmin, float tmax, float &tenter) {
  float tlx = fmaf(bx[0], r.ix, -r.ax), thx = fmaf(bx[3], r.ix, -r.bx);
  float tly = fmaf(bx[1], r.iy, -r.ay), thy = fmaf(bx[4], r.iy, -r.by);
  float tlz = fmaf(bx[2], r.iz, -r.az), thz = fmaf(bx[5], r.iz, -r.bz);
  tenter = fmaxf(fmaxf(fminf(tlx, thx), fminf(tly, thy)), fmaxf(fminf(tlz, thz), tmin));
  float texit = fminf(fminf(fmaxf(tlx, thx), fmaxf(tly, thy)), fminf(fmaxf(tlz, thz), tmax));
  return tenter <= texit * RT_SLACK;
}

template <typename BT> RT_DEV BT tmax_for_boxes(double best_t);
template <> RT_DEV double tmax_for_boxes<double>(double best_t) { return best_t; }
// FP32: rounded up and already widened by the slack, so that the pop-time cull needs no multiplication (the slab test
// then applies the slack twice to this bound: still a superset)
// (round-to-nearest conversion, 0.5 ulp, covered by three more ulps of slack: two instructions; a directed rounding is a
// 13-instruction sequence here, run at every accepted hit)
#ifndef RT_SLACK_UP
#define RT_SLACK_UP 1.0000013f
#endif
template <> RT_DEV float tmax_for_boxes<float>(double best_t) { return (float)best_t * RT_SLACK_UP; }
template <typename BT> RT_DEV BT tmin_for_boxes();
template <> RT_DEV double tmin_for_boxes<double>() { return RT_TMIN; }
template <> RT_DEV float tmin_for_boxes<float>() { return 0.00099999f; }
// pop-time cull: true when a stacked entry distance is provably beyond the closest hit so far
RT_DEV bool cull_entry(float tenter, double tmaxb) { return (double)tenter > tmaxb; }
RT_DEV bool cull_entry(float tenter, float tmaxb) { return tenter > tmaxb; }

// ---------------------------------------------------------------- primitive tests (FP64, reference order of operations)
RT_DEV bool hit_triangle(const DPrim &P, d3 ro, d3 rd, double tmin, double tmax, double &tout) {
  d3 p0 = mk3(P.d[0], P.d[1], P.d[2]), e1 = mk3(P.d[3], P.d[4], P.d[5]), e2 = mk3(P.d[6], P.d[7], P.d[8]);
  d3 h = cross3(rd, e2);                       // Triangle.hs:34
  double a = dot3(e1, h);                      // :35
  if (fabs(a) < 1e-8) return false;            // :36
  double f = 1.0 / a;                          // :39
  d3 s = ro - p0;                              // :40
  double u = f * dot3(s, h);                   // :41
  if (u < 0.0 || u > 1.0) return false;        // :42
  d3 q = cross3(s, e1);                        // :45
  double v = f * dot3(rd, q);                  // :46
  if (v < 0.0 || u + v > 1.0) return false;    // :47
  double t = f * dot3(e2, q);                  // :50
  if (!(t >= tmin && t <= tmax)) return false; // :53 contains
  tout = t;
  return true;
}
RT_DEV bool hit_sphere(const DPrim &P, d3 ro, d3 rd, double tmin, double tmax, double &tout) {
  d3 c = mk3(P.d[0], P.d[1], P.d[2]);
  double r = P.d[3];
  d3 oc = ro - c;                                              // Sphere.hs:29
  double a = dot3(rd, rd);                                     // :30
  double h = dot3(oc, rd);                                     // :31
  double disc = h * h - a * (dot3(oc, oc) - r * r);            // :32
  if (disc < 0) return false;                                  // :25
  double t1 = (-h - sqrt(disc)) / a;                           // :33-34 (near root only)
  if (!(t1 >= tmin && t1 <= tmax)) return false;               // :26
  tout = t1;
  return true;
}
// The reference's slab test returning the whole clipped interval (hitAABBInterval, BoundingBox.hs:18-37)
RT_DEV bool box_interval(const BoxRay<double> &r, const double *bx, double tmin, double tmax, double &tenter, double &texit) {
  double a0, a1, b0, b1, c0, c1;
  const double inf = __builtin_huge_val();
  bool ok = true;
  if (r.px) { ok = ok && !(r.ox < bx[0] || r.ox > bx[3]); a0 = -inf; a1 = inf; }
  else { double t0 = (bx[0] - r.ox) * r.ix, t1 = (bx[3] - r.ox) * r.ix; a0 = fmin(t0, t1); a1 = fmax(t0, t1); }
  if (r.py) { ok = ok && !(r.oy < bx[1] || r.oy > bx[4]); b0 = -inf; b1 = inf; }
  else { double t0 = (bx[1] - r.oy) * r.iy, t1 = (bx[4] - r.oy) * r.iy; b0 = fmin(t0, t1); b1 = fmax(t0, t1); }
  if (r.pz) { ok = ok && !(r.oz < bx[2] || r.oz > bx[5]); c0 = -inf; c1 = inf; }
  else { double t0 = (bx[2] - r.oz) * r.iz, t1 = (bx[5] - r.oz) * r.iz; c0 = fmin(t0, t1); c1 = fmax(t0, t1); }
  tenter = fmax(fmax(a0, b0), fmax(c0, tmin));                     // :30
  texit = fmin(fmin(a1, b1), fmin(c1, tmax));                      // :31
  return ok && tenter <= texit;                                    // :32
}

// Planes are bounded only by the interval their leaf receives (Plane.hs:20 gives every plane the same slab box,
// Plane.hs:26 tests `contains interval t`), so the leaf's FP64 box is re-tested exactly here.  That interval
// is [enter, min(exit, closest so far)] of the leaf's box - except when the leaf is the child visited SECOND and its
// sibling produced a hit: then it is [enter, t firstHit] (BVH.hs:137-138), which can reach past the box's exit.
// Returns 0: miss; 1: hit inside the leaf box's own interval - an ordinary candidate whatever the traversal order;
// 2: the plane is crossed beyond the box's exit but not beyond the closest hit so far: whether the reference sees it
// depends on its traversal history, and the caller settles the whole query with literal_hit below.
#define PLANE_MISS 0
#define PLANE_HIT 1
#define PLANE_LITERAL 2
// A hit at distance t of a primitive classified against the interval its reference leaf receives: planes, which their box does
// NOT contain (Plane.hs:20), and spheres of positive radius, whose exact box c -+ r (Sphere.hs:20-22, no pad) the sphere touches
// in six points - a ray that hits it there has t1 == the box's entry distance in exact arithmetic and one ulp below it in
// FP64 about one time in three, and the reference then sees THROUGH the sphere (`contains interval t1`, Sphere.hs:26).
// bi: index of the leaf's FP64 box in plane_boxes, < 0 for a single-leaf root (the raw ray interval, BVH.hs:121, Trace.hs:24).
// (The leaf's box is a proper box inside all its ancestors' boxes and rounding is monotone, so their clipping adds nothing;
// spheres of radius <= 0, for which that does not hold, go to literal_hit on every candidate hit: see the PRIM step.)
// One evaluation of hitAABBInterval with an open upper end gives te = max(slab entries, tmin) and the box's OWN exit tx;
// the reference enters the leaf iff te <= min(tx, tmax) (tmax = the closest hit so far) and accepts iff te <= t <= tmax -
// with t <= tx an ordinary candidate, beyond it only through [minVal iSecond, t firstHit] (BVH.hs:137-138): literal_hit.
RT_DEV int classify_in_leaf(long long bi, const double *plane_boxes, d3 ro, d3 rd, double tmin, double tmax, double t) {
  if (bi < 0) return (t >= tmin && t <= tmax) ? PLANE_HIT : PLANE_MISS;
  BoxRay<double> br;
  make_boxray(br, ro, rd, 0.0f);
  double te, tx;
  if (!box_interval(br, plane_boxes + 6 * bi, tmin, __builtin_huge_val(), te, tx)) return PLANE_MISS;   // the leaf's box is missed
  if (!(te <= tmax)) return PLANE_MISS;                        // ... or entered only behind a closer hit
  if (!(t >= te && t <= tmax)) return PLANE_MISS;              // `contains interval t`
  return t <= tx ? PLANE_HIT : PLANE_LITERAL;
}
RT_DEV int hit_plane(const DPrim &P, const double *plane_boxes, d3 ro, d3 rd, double tmin, double tmax, double &tout) {
  d3 pp = mk3(P.d[0], P.d[1], P.d[2]), n = mk3(P.d[3], P.d[4], P.d[5]);
  double denom = dot3(rd, n);                                  // Plane.hs:22
  if (!(fabs(denom) > 1e-6)) return PLANE_MISS;                // :23
  double t = dot3(pp - ro, n) / denom;                         // :25
  tout = t;
  return classify_in_leaf((long long)P.d[6], plane_boxes, ro, rd, tmin, tmax, t);      // :26
}

// The reference's `hit` on its own tree (BVH.hs:120-143) walked LITERALLY for one ray: FP64 boxes, every child entered
// with its clipped interval, the second child with [minVal iSecond, t firstHit] when the first one produced a hit
// (:137-138), every primitive tested against the interval its leaf received (Triangle.hs:53, Sphere.hs:26, Plane.hs:26).
// Out of line, per lane, slow, and rare: it runs only for the queries hit_plane reports as PLANE_LITERAL - a plane placed
// outside its constant slab box (Plane.hs:20), where the reference's answer is a property of ITS traversal - and for
// candidate hits on spheres of radius <= 0 (inside-out or point-sized boxes).
// `stk`: this lane's column of 3 * (tree depth) rows of 64 doubles in global memory (ref, lo, hi of the pending second children).
struct LitHit { double t; int best; };
__device__ RT_NOINLINE LitHit literal_hit(const DNode<double> *nodes, const DPrim *prims, int root_ref, double *stk,
                                                        double ox, double oy, double oz, double dx, double dy, double dz,
                                                        double tmin, double tmax) {
  const d3 ro = mk3(ox, oy, oz), rd = mk3(dx, dy, dz);
  BoxRay<double> br;
  make_boxray(br, ro, rd, 0.0f);
  LitHit r;
  r.t = tmax; r.best = -1;
  unsigned long long dirty = 0ull;                   // bit k: some leaf returned a hit since stack entry k was pushed
  int sp = 0, cur = root_ref;
  double lo = tmin, hi = tmax;                       // the root is entered with the raw interval, its own box is never tested
  for (int guard = 0; guard < (1 << 24); ++guard) {
    bool descend = false;
    if (cur >= 0) {                                  // BVHInternal
      const double *bx = nodes[cur].box;
      const int refL = nodes[cur].ref[0], refR = nodes[cur].ref[1];
      double eL, xL, eR, xR;
      const bool hL = box_interval(br, bx, lo, hi, eL, xL);        // :124
      const bool hR = box_interval(br, bx + 6, lo, hi, eR, xR);
      if (hL && hR) {
        const bool lf = eL < eR;                     // :132 (tie => right first)
        stk[(3 * sp) * 64] = (double)(lf ? refR : refL);
        stk[(3 * sp + 1) * 64] = lf ? eR : eL;
        stk[(3 * sp + 2) * 64] = lf ? xR : xL;
        dirty &= ~(1ull << sp);
        ++sp;
        cur = lf ? refL : refR; lo = lf ? eL : eR; hi = lf ? xL : xR;
        descend = true;
      } else if (hL) { cur = refL; lo = eL; hi = xL; descend = true; }   // :126
      else if (hR) { cur = refR; lo = eR; hi = xR; descend = true; }     // :127
    } else {                                         // BVHLeaf (:121-122), closestHitList (HitHelpers.hs:11-16)
      bool have = false;
      double bt = 0.0;
      int bp = -1;
      for (int pi = ~cur;; ++pi) {
        const DPrim P = prims[pi];
        const uint32_t kind = P.kind & DPRIM_KIND_MASK;
        double t = 0.0;
        bool h;
        if (kind == RT_PRIM_TRIANGLE) h = hit_triangle(P, ro, rd, lo, hi, t);
        else if (kind == RT_PRIM_SPHERE) h = hit_sphere(P, ro, rd, lo, hi, t);
        else {
          const double denom = dot3(rd, mk3(P.d[3], P.d[4], P.d[5]));                    // Plane.hs:22
          t = dot3(mk3(P.d[0], P.d[1], P.d[2]) - ro, mk3(P.d[3], P.d[4], P.d[5])) / denom;   // :25
          h = fabs(denom) > 1e-6 && t >= lo && t <= hi;                                  // :23, :26
        }
        if (h && (!have || t <= bt)) { have = true; bt = t; bp = pi; }   // foldr closer: a later element wins an equal t
        if (P.kind & DPRIM_LAST) break;
      }
      if (have) { r.t = bt; r.best = bp; dirty = ~0ull; }   // a second child's hit has t <= t firstHit and replaces it (:141)
    }
    if (descend) continue;
    if (sp == 0) break;                              // back at the root: done
    --sp;
    cur = (int)stk[(3 * sp) * 64]; lo = stk[(3 * sp + 1) * 64]; hi = stk[(3 * sp + 2) * 64];
    if ((dirty >> sp) & 1ull) hi = r.t;              // :137-138: newMaxT = t firstHit replaces maxVal iSecond
  }
  return r;
}

RT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------- the render kernel
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 4
#endif
#include "rt_render_kernel.h"

// ---------------------------------------------------------------- small helper kernels
// One wave per 8x8 tile of the launch (64 pixel slots), one lane per pixel: the pixel's samples summed IN SAMPLE ORDER
// (the reference's foldl', RenderWorker.hs:111-114 - a sequential sum, so no tree reduction), mean, tone map, gamma
// (toneMap / gammaCorrect, Trace.hs:96-105).  A pixel's samples are spp x 24 contiguous bytes, a tile's 64 pixels one
// contiguous region: the wave reads it RS_T samples at a time with all lanes on consecutive addresses (three 192-B runs per
// instruction instead of 64 scattered 8-B reads), stages the sub-block in LDS and each lane then adds its own pixel's run.
// showPixel's quantisation, RenderWorker.hs:108-109: truncate (c * 255.999); the reference neither clamps nor handles NaN
// (its `show` would print the out-of-range integer): clamped to 0..255 here, NaN -> 0
RT_DEV uint8_t quantize8(double c) {
  double v = trunc(c * 255.999);
  v = v != v ? 0.0 : (v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v));
  return (uint8_t)v;
}
#define RS_T 8                      // samples per staged sub-block
#define RS_ROW (3 * RS_T + 1)       // doubles per pixel row in LDS (+1: lanes on different banks)
#define RS_WAVES 4
__global__ __launch_bounds__(64 * RS_WAVES) void k_resolve(const KFrame fr) {
  __shared__ double stage[RS_WAVES][64 * RS_ROW];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t lt = blockIdx.x * RS_WAVES + wv;                      // local tile within the launch
  if (lt >= (uint32_t)(fr.tile_end - fr.tile_begin)) return;          // (whole waves leave: no barrier is used)
  const uint32_t gt = (uint32_t)fr.tile_begin + lt, pos = (uint32_t)lane;
  const int tile = fr.tiles ? fr.tiles[gt] : (int)gt;
  const int px = (tile % fr.tiles_x) * 8 + (int)(pos & 7u), py = (tile / fr.tiles_x) * 8 + (int)(pos >> 3);
  const double *base = fr.samples + 3 * (size_t)lt * 64 * (size_t)fr.spp;   // this tile's 64 runs of 3 spp doubles
  double *st = stage[wv];
  double sx = 0.0, sy = 0.0, sz = 0.0;
  for (int s0 = 0; s0 < fr.spp; s0 += RS_T) {
    const int n = fr.spp - s0 < RS_T ? fr.spp - s0 : RS_T, n3 = 3 * n;
    {                                                                  // lanes 0..47: (pixel 2 it + lane / 24, element lane % 24): two 192-B runs per instruction
      const int pp = lane / (3 * RS_T), k = lane - pp * (3 * RS_T);
      const bool act = pp < 2 && k < n3;
      const double *src = base + ((size_t)pp * fr.spp + s0) * 3 + k;
      double *dst = st + pp * RS_ROW + k;
#pragma unroll 8
      for (int it = 0; it < 32; ++it)
        if (act) dst[2 * it * RS_ROW] = src[(size_t)2 * it * fr.spp * 3];
    }
    __builtin_amdgcn_wave_barrier();                                   // one wave: LDS writes above are visible to its own lanes after the waitcnt the compiler inserts
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const double *r = st + lane * RS_ROW;
    for (int s = 0; s < n; ++s) { sx += r[3 * s]; sy += r[3 * s + 1]; sz += r[3 * s + 2]; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  if (px >= fr.W || py >= fr.H) return;
  const double inv = 1.0 / (double)fr.spp;
  const d3 avg = mk3(inv * sx, inv * sy, inv * sz);                    // RenderWorker.hs:114
  const size_t o = 3 * (fr.packed ? (size_t)gt * 64 + pos : (size_t)py * fr.W + px);
  if (fr.out_radiance) { fr.out_radiance[o] = avg.x; fr.out_radiance[o + 1] = avg.y; fr.out_radiance[o + 2] = avg.z; }
  if (fr.out_color || fr.out_rgb8) {
    const double e = fr.exposure;
    const double tc[3] = {avg.x * e / (avg.x * e + 1), avg.y * e / (avg.y * e + 1), avg.z * e / (avg.z * e + 1)};   // Trace.hs:96-101
    for (int c = 0; c < 3; ++c) {
      const double col = pow_gamma(tc[c], fr.gamma);                                                              // Trace.hs:103-105
      if (fr.out_color) fr.out_color[o + c] = col;
      if (fr.out_rgb8) fr.out_rgb8[o + c] = quantize8(col);                                                       // RenderWorker.hs:108-109
    }
  }
}

// Tiles of tile row y that belong to `rank` of `world`: those with (x + skew * y) mod world == rank, x in [0, tiles_x).
__host__ __device__ inline int tile_first_x(int y, int rank, int world, int skew) {
  return (int)((((long long)rank - (long long)skew * y) % world + world) % world);
}
__host__ __device__ inline int tiles_in_row(int tiles_x, int y, int rank, int world, int skew) {
  const int a = tile_first_x(y, rank, world, skew);
  return a < tiles_x ? (tiles_x - 1 - a) / world + 1 : 0;
}
// A rank's tile list (rt_tile_list's order: increasing y * tiles_x + x), written on the device: one thread per tile row.
__global__ void k_make_tiles(int tiles_x, int tiles_y, int rank, int world, int skew, int32_t *out) {
  const int y = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (y >= tiles_y) return;
  int base = 0;
  for (int q = 0; q < y; ++q) base += tiles_in_row(tiles_x, q, rank, world, skew);
  for (int x = tile_first_x(y, rank, world, skew); x < tiles_x; x += world) out[base++] = y * tiles_x + x;
}

__global__ void k_untile(const int32_t *tiles, int n_tiles, int tiles_x, int W, int H, int n_comp,
                         const double *packed, double *image) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (size_t)n_tiles * 64) return;
  const int tile = tiles[slot >> 6], lane = (int)(slot & 63);
  const int px = (tile % tiles_x) * 8 + (lane & 7), py = (tile / tiles_x) * 8 + (lane >> 3);
  if (px >= W || py >= H) return;
  const size_t dst = ((size_t)py * W + px) * n_comp;
  for (int c = 0; c < n_comp; ++c) image[dst + c] = packed[slot * n_comp + c];
}

// the same scatter for slots of n_bytes bytes each (the 8-bit frame: 3 bytes per pixel)
__global__ void k_untile_bytes(const int32_t *tiles, int n_tiles, int tiles_x, int W, int H, int n_bytes,
                               const uint8_t *packed, uint8_t *image) {
  const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (slot >= (size_t)n_tiles * 64) return;
  const int tile = tiles[slot >> 6], lane = (int)(slot & 63);
  const int px = (tile % tiles_x) * 8 + (lane & 7), py = (tile / tiles_x) * 8 + (lane >> 3);
  if (px >= W || py >= H) return;
  const size_t dst = ((size_t)py * W + px) * n_bytes;
  for (int c = 0; c < n_bytes; ++c) image[dst + c] = packed[slot * n_bytes + c];
}

__global__ void k_quantize(const double *color, uint8_t *rgb8, long long n) {       // showPixel, RenderWorker.hs:108-109
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  rgb8[i] = quantize8(color[i]);
}

// Per (rank, tile row): how many tiles of that rank lie in the rows above - the index of the row's first tile in the rank's list.
__global__ void k_make_row_base(int tiles_x, int tiles_y, int world, int skew, int32_t *row_base) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= world * tiles_y) return;
  const int rank = i / tiles_y, y = i - rank * tiles_y;
  int base = 0;
  for (int q = 0; q < y; ++q) base += tiles_in_row(tiles_x, q, rank, world, skew);
  row_base[i] = base;
}
// The packed tiles of ALL ranks (rank r's slots at packed + r * rank_stride bytes) scattered into the row-major frame in one
// launch: one thread per pixel finds its owner rank, its tile's place in that rank's list and its slot (writes are coalesced).
__global__ void k_untile_all(const int32_t *row_base, int tiles_x, int tiles_y, int world, int skew, int W, int H, int n_bytes,
                             size_t rank_stride, const uint8_t *packed, uint8_t *image) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)W * H) return;
  const int py = (int)(i / (size_t)W), px = (int)(i - (size_t)py * W);
  const int tx = px >> 3, ty = py >> 3;
  const int rank = (int)(((long long)tx + (long long)skew * ty) % world);
  const int lt = row_base[rank * tiles_y + ty] + (tx - tile_first_x(ty, rank, world, skew)) / world;
  const size_t slot = (size_t)lt * 64 + (size_t)((py & 7) * 8 + (px & 7));
  const uint8_t *src = packed + (size_t)rank * rank_stride + slot * n_bytes;
  uint8_t *dst = image + i * n_bytes;
  if ((n_bytes & 7) == 0 && (((size_t)src | (size_t)dst) & 7) == 0) {
    for (int c = 0; c < n_bytes; c += 8) *(unsigned long long *)(dst + c) = *(const unsigned long long *)(src + c);
  } else if ((n_bytes & 3) == 0 && (((size_t)src | (size_t)dst) & 3) == 0) {
    for (int c = 0; c < n_bytes; c += 4) *(uint32_t *)(dst + c) = *(const uint32_t *)(src + c);
  } else {
    for (int c = 0; c < n_bytes; ++c) dst[c] = src[c];
  }
}
