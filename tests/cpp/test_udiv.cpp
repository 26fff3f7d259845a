// fhu::div / fhu::inverse / fhu::inverse_fp (faster_amd/csrc/fh_udiv.hpp) against the integer division they replace: the path search
// splits cell indices (n < 2^27) by a map's row and slice sizes, the decomposition splits voxel numbers (n < 2^28) by the sizes of a
// sub-block (d <= 2^20).  Exhaustive over the small divisors, random and adversarial (multiples of d and their neighbours) beyond.
#include <cstdint>
#include <cstdio>
#include <random>

#include "../../faster_amd/csrc/fh_udiv.hpp"

static long long checks = 0, bad = 0;
static void check(int n, int d, unsigned inv) {
  checks++;
  if (fhu::div(n, d, inv) != n / d && bad++ < 10) std::printf("MISMATCH %d / %d: %d, expected %d\n", n, d, fhu::div(n, d, inv), n / d);
}
static void check_divisor(int d, int n_max, std::mt19937_64& rng, int samples) {
  const unsigned inv = fhu::inverse(d);
  if (d <= (1 << 20)) {
    checks++;
    if (fhu::inverse_fp(d) != inv && bad++ < 10) std::printf("MISMATCH inverse_fp(%d) = %u, expected %u\n", d, fhu::inverse_fp(d), inv);
  }
  for (int n : {0, 1, d - 1, d, d + 1, n_max - 1, n_max - 2, n_max / 2})
    if (n >= 0 && n < n_max) check(n, d, inv);
  for (int k = 0; k < samples; k++) {
    const int n = (int)(rng() % (uint64_t)n_max);
    check(n, d, inv);
    const long long m = (long long)(n / d) * d;  // a multiple of d and its neighbours
    for (long long v : {m - 1, m, m + 1, m + d - 1})
      if (v >= 0 && v < n_max) check((int)v, d, inv);
  }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1;
  std::mt19937_64 rng(99);
  const int n_max = 1 << 28;
  for (int d = 1; d <= (quick ? 20000 : 200000); d++) check_divisor(d, n_max, rng, quick ? 40 : 200);
  for (int e = 1; e <= 27; e++)
    for (int off : {-3, -1, 0, 1, 3}) {
      const int d = (1 << e) + off;
      if (d >= 1) check_divisor(d, n_max, rng, 2000);
    }
  for (int rep = 0; rep < (quick ? 20000 : 400000); rep++) check_divisor(1 + (int)(rng() % (uint64_t)(1 << 20)), n_max, rng, 20);
  for (int rep = 0; rep < (quick ? 5000 : 100000); rep++) check_divisor(1 + (int)(rng() % (uint64_t)(1 << 27)), 1 << 27, rng, 20);
  std::printf("%lld checks, %lld mismatches\n", checks, bad);
  return bad ? 1 : 0;
}
