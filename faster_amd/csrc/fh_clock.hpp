// fh_clock.hpp — the reference's sample clock without running it.
//
// fillX() (/root/reference/faster/src/solverGurobi.cpp:122-168) advances its clock by `t = t + DC` once per sample and moves to the
// next interval when `t > dt_ * (interval + 1)` (:131-135).  The running sum is NOT (i + 1) * DC: every addition rounds, and which
// segment a sample on a knot belongs to depends on it, so the hand-off of a pair (R = sample k of the whole trajectory) has to produce
// the clock's own value.  Running it costs k + 1 dependent double additions on one wavefront — 300 for a typical C4 pair: 25 000
// cycles, more than half of the whole hand-off.  This header computes the same double in ~10 steps:
//
// while t stays inside one binade [2^e, 2^(e+1)) every addition rounds on the same grid u = 2^(e-52): t + DC with DC = q u + r,
// 0 <= r < u, rounds to t + q u if r < u/2 and to t + (q + 1) u if r > u/2 — the SAME increment at every step of the binade (r == u/2,
// round-half-even, depends on the parity of t / u: that binade is walked step by step; it is the one where u/2 is the lowest set bit of
// DC, two steps long for a 53-bit DC).  So a binade is crossed with one multiplication; only the addition that leaves it — its result
// rounds on the coarser grid of the next binade — is performed as the real addition.  All quantities are integers below 2^53 held in
// doubles: every operation below is exact.  Plain C++ (host and device): tests/cpp/test_clock.cpp runs it against the loop.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define FH_CLOCK_FN __host__ __device__ inline
#else
#define FH_CLOCK_FN inline
#endif

namespace fh {

// the loop itself (the definition): t after k + 1 additions, and the interval the reference is in at that sample
FH_CLOCK_FN void clock_loop(int k, double DC, double dt, int N, double& t_out, int& interval_out) {
  double t = 0;
  int interval = 0;
  for (int i = 0; i <= k; i++) {
    t = t + DC;
    if (t > dt * (interval + 1)) interval = (interval + 1 < N - 1) ? interval + 1 : N - 1;
  }
  t_out = t;
  interval_out = interval;
}

FH_CLOCK_FN void clock_at(int k, double DC, double dt, int N, double& t_out, int& interval_out) {
  // the short cut needs: a positive normal DC far from overflow, and at most one knot per step (dt >= 2 DC — findDT never goes below
  // 2 DC, solverGurobi.cpp:494-497 — so that "one interval per step at most" never lags behind the knots)
  if (!(DC >= 1e-280 && DC <= 1e280 && dt >= 2.0 * DC && k >= 0 && k < (1 << 30))) {
    clock_loop(k, DC, dt, N, t_out, interval_out);
    return;
  }
  double t = 0;
  int m = k + 1;  // additions still to make
  {  // the first binades hold a step or two each: walked plainly
    const int warm = m < 16 ? m : 16;
    for (int i = 0; i < warm; i++) t = t + DC;
    m -= warm;
  }
  const double B = 9007199254740992.0;  // 2^53
  while (m > 0) {
    int e;
    (void)frexp(t, &e);                       // t in [2^(e-1), 2^e): grid u = 2^(e-53)
    const double su = ldexp(1.0, 53 - e);     // 1 / u
    const double T = t * su;                  // t / u: an integer in [2^52, 2^53)
    const double Ds = DC * su;                // DC / u (exact: a power of two)
    const double q = floor(Ds), fr = Ds - q;
    if (fr == 0.5) {                          // round-half-even binade: one real addition at a time
      t = t + DC;
      m--;
      continue;
    }
    const double I = q + (fr > 0.5 ? 1.0 : 0.0);  // what one addition adds, in units of u, while the sum stays below 2^e
    const double room = B - 1.0 - q - T;          // additions j = 0, 1, ... stay on this grid while T + j I + q <= 2^53 - 1
    double j = 0.0;
    if (room >= 0.0 && I > 0.0) {
      double j0 = floor(room / I);
      if (j0 * I > room) j0 -= 1.0;               // (the quotient may have rounded across an integer)
      else if ((j0 + 1.0) * I <= room) j0 += 1.0;
      j = j0 + 1.0;
      if (j > (double)m) j = (double)m;
    }
    if (j > 0.0) {
      t = ldexp(T + j * I, e - 53);
      m -= (int)j;
    }
    if (m > 0) {                                  // the addition that leaves the binade (or a step that adds less than u: I == 0)
      t = t + DC;
      m--;
    }
  }
  int interval = 0;
  for (int jn = 1; jn < N; jn++)
    if (t > dt * jn) interval = jn;               // (monotone clock, at most one knot per step: the knots passed are the knots below t)
  t_out = t;
  interval_out = interval;
}

}  // namespace fh
