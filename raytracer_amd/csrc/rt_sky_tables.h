// Host side of the sky lookup (sky_texel, rt_kernels.h): the tables that let the device pick the texel the REFERENCE's
// floating-point pipeline picks (SkySphere.hs:35-41),
//     i = clamp(floor((0.5 + atan2 z x / (2 pi)) w)),   j = clamp(floor((0.5 - asin y / pi) h)),
// without evaluating atan2 / asin per path, INCLUDING for directions within rounding of a texel boundary.  A camera that looks
// along an axis puts a whole column of rays there, and which texel such a ray gets is decided by how each operation rounds.
//
// Rows.  j is a monotone step function of the DOUBLE y, so each boundary k has a largest double T_k with j(y) >= k: it is
// found by bisection over the doubles, evaluating the reference's expression with the platform's asin (GHC's asin on Double is
// libm's).  `y <= T_k` on the device then IS the reference's row for every y - there is no near-boundary case.
//
// Columns.  `atan2` on Double is NOT libm's atan2 under GHC: instance RealFloat Double does not define it, so the class default
// of GHC.Float (the Haskell 2010 report's definition, section 9 Prelude) runs:
//     atan2 y x | x > 0 = atan (y/x) | x == 0 && y > 0 = pi/2 | x < 0 && y > 0 = pi + atan (y/x)
//               | (x <= 0 && y < 0) || (x < 0 && isNegativeZero y) || (both negative zero) = -atan2 (-y) x
//               | y == 0 && (x < 0 || isNegativeZero x) = pi | x == 0 && y == 0 = y | otherwise = x + y
// so within each of the three regions  x > 0,  x < 0 < z,  x < 0 and z < 0  the column is a monotone step function of the
// DOUBLE quotient q = z / x (IEEE division: the same double on the host, on the device and under GHC), and each boundary k
// has a smallest double Q_k with i(q) >= k there - bisection again, with the platform's atan.  The device's cross-product
// test settles every direction that is not within rounding of a boundary; for the few that are it forms q and compares it
// with Q_k (side_q).  Zeros of x or z (the other guards above) agree with libm's atan2 and take the device's libm path.
//
// Host only (but for side_q); no counterpart in the reference, which simply calls atan2 / asin.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#ifdef __HIPCC__
#define SKYTAB_HD __host__ __device__
#else
#define SKYTAB_HD
#endif

namespace skytab {

// The device's test for a direction within rounding of column boundary k (sky_texel_exact, rt_kernels.h; here so that a host test runs
// the same lines): does the reference put (x, z) in a column >= k?  Q = this boundary's thresholds on z / x for the regions
// x > 0, x < 0 < z, x < 0 and z < 0.  1 / 0, or -1 when the question is not one for this test (a zero, a NaN, no table entry).
SKYTAB_HD inline int side_q(const double *Q, double x, double z) {
  if (x == 0.0 || z == 0.0) return -1;
  const double q = z / x;
  const double t = x > 0.0 ? Q[0] : (z > 0.0 ? Q[1] : Q[2]);
  if (!(t == t) || !(q == q)) return -1;
  return q >= t ? 1 : 0;
}

// the doubles as an ordered set of integers (monotone bijection, -0 and +0 adjacent)
inline int64_t ord_of(double v) { int64_t b; std::memcpy(&b, &v, 8); return b < 0 ? (int64_t)0x8000000000000000ull - b : b; }
inline double of_ord(int64_t o) { int64_t b = o < 0 ? (int64_t)0x8000000000000000ull - o : o; double v; std::memcpy(&v, &b, 8); return v; }

// the reference's expressions (SkySphere.hs:35-36,39-40), before the clamp.  region 0: x > 0, 1: x < 0 < z, 2: x < 0 and z < 0;
// q = z / x (in region 2 the reference forms (-z) / x = -q and negates the result)
inline double hs_atan2_of_quotient(double q, int region) {
  const double pi = 3.14159265358979323846;
  return region == 0 ? std::atan(q) : (region == 1 ? pi + std::atan(q) : -(pi + std::atan(-q)));
}
inline double col_of_quotient(double q, int region, int w) { return std::floor((0.5 + hs_atan2_of_quotient(q, region) / (2 * 3.14159265358979323846)) * (double)w); }
inline double row_of_y(double y, int h) { return std::floor((0.5 - std::asin(y) / 3.14159265358979323846) * (double)h); }

// largest double y in [-1, 1] with row_of_y(y) >= k   (1 <= k <= h; row_of_y(-1) = h, row_of_y(1) = 0)
inline double row_threshold(int k, int h) {
  int64_t lo = ord_of(-1.0), hi = ord_of(1.0);    // row(lo) >= k, row(hi) < k
  while ((uint64_t)hi - (uint64_t)lo > 1) {        // (the difference can exceed INT64_MAX)
    const int64_t mid = lo + (int64_t)(((uint64_t)hi - (uint64_t)lo) / 2);
    if (row_of_y(of_ord(mid), h) >= (double)k) lo = mid; else hi = mid;
  }
  return of_ord(lo);
}
// smallest double q with col_of_quotient(q, region) >= k; -inf if every q has it, +inf if none below +inf has
inline double col_threshold(int k, int region, int w) {
  const double inf = std::numeric_limits<double>::infinity();
  if (col_of_quotient(-inf, region, w) >= (double)k) return -inf;
  if (!(col_of_quotient(inf, region, w) >= (double)k)) return inf;
  int64_t lo = ord_of(-inf), hi = ord_of(inf);
  while ((uint64_t)hi - (uint64_t)lo > 1) {
    const int64_t mid = lo + (int64_t)(((uint64_t)hi - (uint64_t)lo) / 2);
    if (col_of_quotient(of_ord(mid), region, w) >= (double)k) hi = mid; else lo = mid;
  }
  return of_ord(hi);
}

// pad: the tables start at boundary -pad and end at boundary n + pad.
//   u_tab: (cos phi_k, sin phi_k) in double, phi_k = 2 pi (k / w - 0.5): the columns' nominal boundary directions
//   u_q:   Q_k for the three regions, 1 <= k <= w - 1; NaN elsewhere (the +-pi cut, where the column saturates: see sky_texel)
//   v_tab: T_k for 1 <= k <= h, +inf for k <= 0 (every y is in a row >= 0), -inf for k > h
inline void build(int w, int h, int pad, std::vector<double> &u_tab, std::vector<double> &u_q, std::vector<double> &v_tab) {
  const double pi = 3.14159265358979323846, nan = std::numeric_limits<double>::quiet_NaN();
  u_tab.assign(2 * (size_t)(w + 2 * pad + 1), 0.0);
  u_q.assign(3 * (size_t)(w + 2 * pad + 1), nan);
  for (int k = -pad; k <= w + pad; ++k) {
    const size_t q = (size_t)(k + pad);
    const double phi = 2 * pi * ((double)k / (double)w - 0.5);
    u_tab[2 * q] = std::cos(phi);
    u_tab[2 * q + 1] = std::sin(phi);
    if (k >= 1 && k <= w - 1)
      for (int r = 0; r < 3; ++r) u_q[3 * q + r] = col_threshold(k, r, w);
  }
  v_tab.assign((size_t)(h + 2 * pad + 1), 0.0);
  for (int k = -pad; k <= h + pad; ++k)
    v_tab[(size_t)(k + pad)] = k <= 0 ? INFINITY : (k > h ? -INFINITY : row_threshold(k, h));
}

}  // namespace skytab
