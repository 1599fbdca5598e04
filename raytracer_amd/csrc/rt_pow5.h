// x ** 5 as the reference computes it (schlick, Trace.hs:85-89: `(1 - cosine) ** 5`; (**) on Double is libm's pow).  The result
// is compared with a random number (Trace.hs:54-57: reflect or refract), so its last bit can send a whole path another way:
// x*x*x*x*x differs from pow(x, 5) in the last bit for 41 % of all x, a CORRECTLY ROUNDED x^5 for 0.09 % (glibc 2.35's pow is
// within 0.52 ulp; 300,000 random x, tests/test_pow5.py).  So: x^2 and x^4 in double-double (exact products by fma, ~1e-31
// relative), one final rounding.  Outside 1e-50 < |x| < 1e50 (zero, infinities, NaN, results that underflow) the plain product:
// the same special values as pow, and nothing there can be within an ulp of a random number's 2^-32 grid.
// Shared by the device (rt_render_kernel.h) and a host test.
#pragma once
#include <cmath>
#ifdef __HIPCC__
#define RT_POW5_HD __host__ __device__
#else
#define RT_POW5_HD
#endif

RT_POW5_HD inline double pow5_cr(double x) {
  const double ax = fabs(x);
  if (!(ax > 1e-50 && ax < 1e50)) return x * x * x * x * x;
  const double p = x * x, pe = fma(x, x, -p);                 // x^2 = p + pe exactly
  const double q = p * p, qe = fma(p, p, -q) + 2.0 * (p * pe);   // x^4 = q + qe (+ pe^2, 1e-32 relative)
  const double r = q * x, re = fma(q, x, -r) + qe * x;        // x^5 = r + re
  return r + re;
}
