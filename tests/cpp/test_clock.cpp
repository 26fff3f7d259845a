// fh::clock_at (faster_amd/csrc/fh_clock.hpp) against the loop it replaces — the reference's `t = t + DC` clock of fillX
// (/root/reference/faster/src/solverGurobi.cpp:131-135): bit-identical t and the same interval for every sample index tried.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "../../faster_amd/csrc/fh_clock.hpp"

static long long checks = 0, bad = 0;
static void check(int k, double DC, double dt, int N) {
  double t0, t1;
  int i0, i1;
  fh::clock_loop(k, DC, dt, N, t0, i0);
  fh::clock_at(k, DC, dt, N, t1, i1);
  checks++;
  if (std::memcmp(&t0, &t1, 8) != 0 || i0 != i1) {
    if (bad++ < 10) std::printf("MISMATCH k=%d DC=%.17g dt=%.17g N=%d: loop (%.17g, %d) short cut (%.17g, %d)\n", k, DC, dt, N, t0, i0, t1, i1);
  }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1;  // (the CPU suite; the full run takes half a minute)
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0.0, 1.0);
  // the reference's DC and its neighbours, every k of a long trajectory, knots exactly on samples and just beside them
  const double dcs[] = {0.01, 0.01000000000000000194, 0.02, 0.005, 0.0078125, 0.1, 1.0 / 3.0, 0.25, 1e-3, 0.0123456789, 0.05, 3.0, 0.3};
  for (double DC : dcs)
    for (int N : {1, 2, 6, 10, 15, 16})
      for (double mult : {2.0, 2.5, 7.0, 24.494897, 50.0, 73.48469495773315, 100.0}) {
        const double dt = DC * mult;
        for (int k = 0; k < (quick ? 1200 : 3000); k++) check(k, DC, dt, N);
      }
  // random step sizes with random mantissa lengths (the tie binade moves with the lowest set bit of DC), random sample indices
  for (int rep = 0; rep < (quick ? 20000 : 200000); rep++) {
    uint64_t bits;
    double DC = std::ldexp(0.5 + 0.5 * U(rng), (int)(rng() % 24) - 12);
    std::memcpy(&bits, &DC, 8);
    const int keep = 1 + (int)(rng() % 52);  // mantissa bits kept
    bits &= ~((1ull << (52 - keep)) - 1ull);
    std::memcpy(&DC, &bits, 8);
    const int N = 1 + (int)(rng() % 16);
    const double dt = DC * (2.0 + 200.0 * U(rng));
    const int k = (int)(rng() % 20000);
    check(k, DC, dt, N);
    if (rep % 16 == 0) check((int)(rng() % 400000), DC, dt, N);
  }
  // outside the short cut's conditions it falls back to the loop: still identical
  check(100, 0.01, 0.015, 10);
  check(100, 0.0, 1.0, 4);
  check(0, 0.01, 0.5, 10);
  std::printf("%lld checks, %lld mismatches\n", checks, bad);
  return bad ? 1 : 0;
}
