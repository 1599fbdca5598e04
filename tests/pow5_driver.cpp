// Test driver for raytracer_amd/csrc/rt_pow5.h (tests/test_pow5.py compiles it with g++ -ffp-contract=off): prints pow5_cr(x) and
// libm's pow(x, 5) for the doubles read from stdin (hex floats), one pair per line.
#include "rt_pow5.h"
#include <cstdio>
#include <cstdlib>
int main() {
  char buf[128];
  while (fgets(buf, sizeof buf, stdin)) { const double x = strtod(buf, nullptr); printf("%a %a\n", pow5_cr(x), std::pow(x, 5.0)); }
}
