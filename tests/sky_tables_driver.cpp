// Test driver for raytracer_amd/csrc/rt_sky_tables.h (tests/test_sky_tables.py compiles it with g++): prints the row thresholds of a
// w x h sky and, for n directions within 1e-19 ... 1e-14 rad of a column boundary, what the device's side test (skytab::side_q) says.
#include "rt_sky_tables.h"
#include <cstdio>
#include <cstdlib>
#include <random>
int main(int argc, char **argv) {
  int w = atoi(argv[1]), h = atoi(argv[2]), n = atoi(argv[3]);
  std::vector<double> ut, uq, vt;
  skytab::build(w, h, 2, ut, uq, vt);
  for (int k = -2; k <= h + 2; ++k) printf("R %d %a\n", k, vt[k + 2]);
  std::mt19937_64 g(7);
  std::uniform_real_distribution<double> U(0, 1);
  for (int it = 0; it < n; ++it) {
    int k = 1 + (int)(U(g) * (w - 1)); if (k > w - 1) k = w - 1;
    if (w < 2) break;
    double phi = 2 * 3.14159265358979323846 * ((double)k / w - 0.5);
    double delta = (U(g) < 0.5 ? -1 : 1) * std::pow(10.0, -19 + 5 * U(g));
    double r = std::pow(10.0, -3 * U(g));
    double x = r * std::cos(phi + delta), z = r * std::sin(phi + delta);
    x = std::nextafter(x, U(g) < 0.5 ? -10 : 10); if (U(g) < 0.3) z = std::nextafter(z, U(g) < 0.5 ? -10 : 10);
    printf("C %d %a %a %d\n", k, x, z, skytab::side_q(&uq[3 * (k + 2)], x, z));
  }
}
