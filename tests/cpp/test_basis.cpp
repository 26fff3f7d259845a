// Checks the reduced-space basis tables (faster_amd/csrc/fh_basis.hip.hpp) on the host: orthogonality, that the complement
// columns annihilate the final-state functionals of setConstraintsXf (/root/reference/faster/src/solverGurobi.cpp:332-357) at
// any step h, that xp = Q[:, :3] (M rho) is the minimum-norm solution of the equalities, the consistency rows for N < 3, and
// the inverse row norms against a direct computation at a random h.  Prints "ok <checks>" or the first failure.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../faster_amd/csrc/fh_basis.hip.hpp"

static int fails = 0, checks = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    checks++;                                             \
    if (!(cond)) {                                        \
      if (fails < 20) { std::printf("FAIL %s: ", #cond); std::printf(__VA_ARGS__); std::printf("\n"); } \
      fails++;                                            \
    }                                                     \
  } while (0)

int main() {
  const std::vector<double> tab = fh::build_basis_tables();
  unsigned long long rng = 88172645463325252ull;
  auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (double)(rng >> 11) / 9007199254740992.0; };
  for (int N = 1; N <= FH_MAX_SEG; N++) {
    const double* T = tab.data() + (size_t)(N - 1) * fh::BT_STRIDE;
    const double* Q = T + fh::BT_Z;
    for (int a = 0; a < N; a++)
      for (int b = 0; b < N; b++) {
        double d = 0;
        for (int s = 0; s < N; s++) d += Q[s * N + a] * Q[s * N + b];
        CHECK(std::fabs(d - (a == b ? 1.0 : 0.0)) < 1e-14, "N %d columns %d %d: %g", N, a, b, d);
      }
    for (int ff = 0; ff < 2; ff++) {
      const int nrow = ff ? 3 : 2, koff = 3 - nrow, c0 = 2 + ff;
      const double* E = T + fh::BT_EQ + ff * fh::BT_EQ_WORDS;
      const int mask = (int)E[18];
      for (int rep = 0; rep < 4; rep++) {
        const double h = 0.05 + 3.0 * rnd();
        const int wk[3] = {0, 3, 4};  // W_P, W_V, W_A
        const double hp[3] = {h * h * h, h * h, h};
        // complement columns annihilate the functionals
        for (int j = 0; j < nrow; j++)
          for (int c = c0; c < N; c++) {
            double d = 0, w2 = 0;
            for (int s = 0; s < N; s++) { const double w = fh::basis_wcoef(wk[j + koff], N - 1 - s, h); d += w * Q[s * N + c]; w2 += w * w; }
            CHECK(std::fabs(d) <= 1e-13 * std::sqrt(w2), "N %d ff %d row %d column %d: %g", N, ff, j, c, d);
          }
        // xp from a random right-hand side
        double rhs[3], rho[3], cl[3] = {0, 0, 0}, xp[FH_MAX_SEG];
        for (int j = 0; j < nrow; j++) { rhs[j] = 4.0 * rnd() - 2.0; rho[j] = rhs[j] / hp[j + koff]; }
        for (int j = nrow; j < 3; j++) rho[j] = rhs[j] = 0.0;
        for (int l = 0; l < 3; l++)
          for (int j = 0; j < 3; j++) cl[l] += E[l * 3 + j] * rho[j];
        for (int s = 0; s < N; s++) {
          xp[s] = 0;
          for (int l = 0; l < 3 && l < N; l++) xp[s] += Q[s * N + l] * cl[l];
        }
        for (int j = 0; j < nrow; j++) {
          double d = 0, w2 = 0;
          for (int s = 0; s < N; s++) { const double w = fh::basis_wcoef(wk[j + koff], N - 1 - s, h); d += w * xp[s]; w2 += w * w; }
          if (!((mask >> j) & 1)) CHECK(std::fabs(d - rhs[j]) <= 1e-11 * (1.0 + std::fabs(rhs[j])), "N %d ff %d h %g row %d: %g vs %g", N, ff, h, j, d, rhs[j]);
          else {  // dependent row: the table's residual is what the row misses by
            double g = 0;
            for (int c = 0; c < 3; c++) g += E[9 + j * 3 + c] * rho[c];
            CHECK(std::fabs(hp[j + koff] * g - (rhs[j] - d)) <= 1e-10 * (1.0 + std::fabs(rhs[j]) + std::fabs(d)), "N %d ff %d residual row %d: %g vs %g", N, ff, j,
                  hp[j + koff] * g, rhs[j] - d);
          }
        }
        CHECK(mask == 0 || N < nrow, "N %d ff %d mask %d", N, ff, mask);
        if (N < nrow) CHECK(mask == ((1 << nrow) - 1) - ((1 << N) - 1), "N %d ff %d mask %d", N, ff, mask);
        for (int c = c0; c < N; c++) {  // minimum norm: no component in the complement
          double d = 0;
          for (int s = 0; s < N; s++) d += xp[s] * Q[s * N + c];
          CHECK(std::fabs(d) < 1e-12 * (1.0 + std::fabs(cl[0]) + std::fabs(cl[1]) + std::fabs(cl[2])), "N %d ff %d xp column %d: %g", N, ff, c, d);
        }
        // inverse row norms
        const double* C = T + fh::BT_C + ff * fh::BT_C_WORDS;
        for (int kind = 0; kind < 5; kind++)
          for (int tt = 0; tt <= N; tt++) {
            double o2 = 0, w2 = 0;
            for (int s = 0; s < tt && s < N; s++) { const double w = fh::basis_wcoef(kind, tt - 1 - s, h); w2 += w * w; }
            for (int c = c0; c < N; c++) {
              double o = 0;
              for (int s = 0; s < tt && s < N; s++) o += Q[s * N + c] * fh::basis_wcoef(kind, tt - 1 - s, h);
              o2 += o * o;
            }
            const double scale = kind == 3 ? h * h : (kind == 4 ? h : h * h * h);
            const double inv = C[kind * fh::BT_C_TT + tt] / scale;
            if (o2 > 1e-18 * w2 && o2 > 0) CHECK(std::fabs(inv * std::sqrt(o2) - 1.0) < 1e-9, "N %d ff %d kind %d tt %d: %g", N, ff, kind, tt, inv * std::sqrt(o2));
            else CHECK(inv == 0.0, "N %d ff %d kind %d tt %d: constant row has weight %g", N, ff, kind, tt, inv);
          }
        // structurally constant rows of a whole trajectory: control points 1..3 of the last segment
        if (ff && N >= 4) {
          CHECK(C[0 * fh::BT_C_TT + N] == 0.0 && C[1 * fh::BT_C_TT + N - 1] == 0.0 && C[2 * fh::BT_C_TT + N - 1] == 0.0, "N %d: last-segment control points", N);
          CHECK(C[0 * fh::BT_C_TT + N - 1] > 0.0, "N %d: control point 0 of the last segment is free", N);
        }
      }
    }
  }
  if (fails) { std::printf("%d of %d checks failed\n", fails, checks); return 1; }
  std::printf("ok %d\n", checks);
  return 0;
}
