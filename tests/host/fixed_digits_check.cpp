// fixed_digits (csrc/report.cpp) against snprintf on random and adversarial doubles:
//   g++ -O2 -std=c++17 -pthread -o /tmp/fdc tests/host/fixed_digits_check.cpp && /tmp/fdc
#include "../../km_amd/csrc/report.cpp"
#include <random>
int main() {
  std::mt19937_64 rng(12345);
  long bad = 0, n = 0;
  auto check = [&](double v) {
    for (int d : {1, 3}) {
      char a[64], b[64];
      const int na = fixed_digits(v, d, a);
      const int nb = snprintf(b, sizeof b, d == 3 ? "%.3f" : "%.1f", v);
      ++n;
      if (na == 0) continue;
      if (na != nb || memcmp(a, b, (size_t)na)) { if (bad++ < 10) fprintf(stderr, "%.17g: %.*s vs %s\n", v, na, a, b); }
    }
  };
  for (int i = 0; i < 4000000; ++i) {
    uint64_t bits = rng();
    double v; memcpy(&v, &bits, 8);
    check(v);                                                   // any bit pattern
    check((double)(int64_t)(rng() % 2000001 - 1000000) / 2000.0);   // ties of %.3f
    check((double)(int64_t)(rng() % 2000001 - 1000000) / 20.0);     // ties of %.1f
    check((double)(rng() % 100000) / (double)(1 + rng() % 1000));   // ratios of small integers
    check(std::ldexp((double)(rng() >> 11), -(int)(rng() % 120)));
  }
  for (double v : {0.0, -0.0, 0.0005, 0.00049999999999999999, 0.05, 0.25, 0.35, 1e-320, 4.9e-324, 1e15, 1.1258999e15, 0.9995, 0.99949999999999994, -0.0004, 2.5, 3.5, 1e14 + 0.05})
    check(v);
  printf("%ld checks, %ld mismatches\n", n, bad);
  return bad != 0;
}
