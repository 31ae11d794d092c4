// Host check of threshold_shortcut (km_amd/csrc/device_common.h): for every parameter pair, every
// sum below the bound it returns must have the threshold it returns — decided by the very float64
// expression the kernels fall back on (child_threshold, Jellyfish.get_child's comparison,
// km/utils/Jellyfish.py:69-72) — and the bound must be tight.  Prints "ok <cases>" or the first
// counter-example.  No GPU is touched.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <random>
#include "../../km_amd/csrc/device_common.h"

using namespace kmd;

int main() {
  std::mt19937_64 rng(20261004);
  const double ratios[] = {0.0, 1e-9, 0.001, 0.01, 0.05, 0.05000000000000001, 0.1, 0.25, 0.3, 1.0 / 3.0, 0.5, 0.999, 1.0, 2.5, -0.05};
  const long long cutoffs[] = {-3, 0, 1, 2, 5, 7, 50, 500, 65534, 65535, 70000, 4294967295LL, 4294967296LL};
  long cases = 0;
  auto check = [&](double ratio, long long nc) -> bool {
    uint64_t below; uint32_t T;
    threshold_shortcut(ratio, nc, &below, &T);
    // sample sums densely near 0 and near the bound, sparsely in between
    auto same = [&](uint64_t sum) {
      bool none; const uint32_t t = child_threshold(sum, ratio, nc, &none);
      return !none && t == T;
    };
    for (uint64_t s = 0; s < below && s < 3000; ++s) if (!same(s)) { printf("FAIL ratio %.17g cutoff %lld sum %llu\n", ratio, nc, (unsigned long long)s); return false; }
    for (uint64_t s = below > 3000 ? below - 3000 : 0; s < below; ++s) if (!same(s)) { printf("FAIL ratio %.17g cutoff %lld sum %llu\n", ratio, nc, (unsigned long long)s); return false; }
    for (int i = 0; i < 2000 && below > 6000; ++i) { const uint64_t s = rng() % below; if (!same(s)) { printf("FAIL ratio %.17g cutoff %lld sum %llu\n", ratio, nc, (unsigned long long)s); return false; } }
    // tight: the first sum past the bound (if it is within the range sums can take) decides otherwise,
    // or the product there exceeds the cut-off
    if (below > 0 && below < (1ull << 36)) {
      if ((double)below * ratio <= (double)nc) { printf("LOOSE ratio %.17g cutoff %lld below %llu\n", ratio, nc, (unsigned long long)below); return false; }
    }
    ++cases;
    return true;
  };
  for (double r : ratios) for (long long c : cutoffs) if (!check(r, c)) return 1;
  std::uniform_real_distribution<double> ur(0.0, 1.0);
  for (int i = 0; i < 3000; ++i) {
    const double r = ur(rng) < 0.5 ? ur(rng) : ur(rng) * 0.1;
    const long long c = (long long)(rng() % 200000) - 5;
    if (!check(r, c)) return 1;
  }
  printf("ok %ld\n", cases);
  return 0;
}
