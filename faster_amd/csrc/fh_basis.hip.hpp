// fh_basis.hip.hpp — the reduced-space basis tables of the solve kernels (computed once per context on the host, read-only on
// the device).
//
// The final-state equalities of a trajectory (setConstraintsXf, /root/reference/faster/src/solverGurobi.cpp:332-357: velocity
// and acceleration at the end of the last segment, and the position for the whole trajectory) are, per axis, 2 or 3 linear
// functionals of the N jerks: w_A = h, w_V = h^2 (1/2 + m), w_P = h^3 (1/6 + m/2 + m^2/2) with m = N-1-s.  Whatever the step h,
// they span the polynomials of degree < 2 (safe) or < 3 (whole) in s on the N points s = 0..N-1.  The kernels therefore solve
// every QP of a problem in the orthogonal complement of those polynomials: jerks x = xp + (Z (x) I3) y with Z an orthonormal
// basis of the complement (N x (N-2) or N x (N-3)), xp the minimum-norm solution of the equalities, cost |xp|^2 + |y|^2 — 21 or
// 24 unknowns instead of 30 at N = 10 and no equality rows in the factorisation.  Z depends on N only (the discrete orthogonal
// polynomials of degree >= 2 / >= 3, up to a rotation): it is a mathematical constant, tabulated here for N = 1..FH_MAX_SEG.
//
// Per N (BT_STRIDE doubles):
//   BT_Z    Q[N][N]   orthogonal, row major (row = segment s): columns 0,1,2 = orthonormal polynomials of degree 0,1,2 (Householder
//                     QR of [1, s, s^2]), columns 3.. = complement.  Safe problems use columns 2..N-1, whole problems 3..N-1.
//   BT_EQ   per force_final (0 safe, 1 whole), 24 doubles: M[3][3], G[3][3], accepted-row mask, pad.
//                     rho_j = (target_j - zero-jerk end state_j) / h^p_j for the rows j in the reference's order ([pos], vel, acc;
//                     p = 3, 2, 1); xp = sum_l Q[:, l] c_l with c = M rho; a row that is linearly dependent on the rows before it
//                     (N < 3 only) is consistent iff |h^p_j (G rho)_j| <= feas_tol.  (Gram-Schmidt in the reference's row order at
//                     h = 1, as the CPU oracle does it per trial: directions do not depend on h, lengths scale with h^p.)
//   BT_C    per force_final: C[5 kinds][17]: 1 / |Z^T w(kind, tt)| at h = 1 for the row kinds W_P, W_CP1, W_CP2, W_V, W_A of the
//                     state at the start of segment tt (0: the row does not depend on y — a constant row).  Scales with h^-3
//                     (P, CP1, CP2), h^-2 (V), h^-1 (A).
//   BT_CJ   per force_final: CJ[16]: 1 / |Z[t, :]| for the jerk box rows of segment t.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/fasterhip.h"

namespace fh {

enum { BT_Z = 0, BT_EQ = 256, BT_EQ_WORDS = 24, BT_C = 304, BT_C_TT = 17, BT_C_WORDS = 5 * 17, BT_CJ = 474, BT_CJ_WORDS = 16,
       BT_STRIDE = 512 };
static_assert(FH_MAX_SEG <= 16, "table layout");
static_assert(BT_EQ + 2 * BT_EQ_WORDS == BT_C && BT_C + 2 * BT_C_WORDS == BT_CJ && BT_CJ + 2 * BT_CJ_WORDS <= BT_STRIDE, "table layout");

// coefficient of jerk j_s in functional `kind` (0 P, 1 CP1, 2 CP2, 3 V, 4 A) of the state m+1 segments later, step h
inline double basis_wcoef(int kind, int m, double h) {
  const double dm = (double)m;
  const double cP = h * h * h * (1.0 / 6.0 + 0.5 * dm + 0.5 * dm * dm);
  const double cV = h * h * (0.5 + dm);
  switch (kind) {
    case 0: return cP;
    case 1: return cP + cV * (h / 3.0);
    case 2: return cP + cV * (2.0 * h / 3.0) + h * (h * h / 6.0);
    case 3: return cV;
    default: return h;
  }
}

// table[(N - 1) * BT_STRIDE + ...], N = 1..FH_MAX_SEG
inline std::vector<double> build_basis_tables() {
  std::vector<double> tab((size_t)FH_MAX_SEG * BT_STRIDE, 0.0);
  const double dep2 = 1e-20;  // (fh_default_params' dep_tol squared; accepted rows are far from the threshold: see header)
  for (int N = 1; N <= FH_MAX_SEG; N++) {
    double* T = tab.data() + (size_t)(N - 1) * BT_STRIDE;
    // ---- Q = H0 H1 H2 from the Householder QR of V = [1, s, s^2] (N x 3) ----
    const int nref = N < 3 ? N : 3;
    std::vector<double> V((size_t)N * 3), hv((size_t)3 * N, 0.0);
    for (int s = 0; s < N; s++) { V[s * 3 + 0] = 1.0; V[s * 3 + 1] = (double)s; V[s * 3 + 2] = (double)s * (double)s; }
    for (int j = 0; j < nref; j++) {
      double nrm = 0.0;
      for (int s = j; s < N; s++) nrm += V[s * 3 + j] * V[s * 3 + j];
      nrm = std::sqrt(nrm);
      double* v = &hv[(size_t)j * N];
      for (int s = j; s < N; s++) v[s] = V[s * 3 + j];
      v[j] += (V[j * 3 + j] >= 0.0 ? nrm : -nrm);
      double vv = 0.0;
      for (int s = j; s < N; s++) vv += v[s] * v[s];
      if (vv > 0.0) {
        const double inv = 1.0 / std::sqrt(vv);
        for (int s = j; s < N; s++) v[s] *= inv;
        for (int c = j; c < 3; c++) {
          double dot = 0.0;
          for (int s = j; s < N; s++) dot += v[s] * V[s * 3 + c];
          for (int s = j; s < N; s++) V[s * 3 + c] -= 2.0 * dot * v[s];
        }
      }
    }
    std::vector<double> Q((size_t)N * N, 0.0);
    for (int c = 0; c < N; c++) {
      std::vector<double> e((size_t)N, 0.0);
      e[c] = 1.0;
      for (int j = nref - 1; j >= 0; j--) {
        const double* v = &hv[(size_t)j * N];
        double dot = 0.0;
        for (int s = 0; s < N; s++) dot += v[s] * e[s];
        for (int s = 0; s < N; s++) e[s] -= 2.0 * dot * v[s];
      }
      for (int s = 0; s < N; s++) Q[(size_t)s * N + c] = e[s];
    }
    for (int i = 0; i < N * N; i++) T[BT_Z + i] = Q[i];
    for (int ff = 0; ff < 2; ff++) {
      const int nrow = ff ? 3 : 2, koff = 3 - nrow;  // row j: kind j + koff (0 pos, 1 vel, 2 acc)
      // ---- Gram-Schmidt of the equality functionals at h = 1, in the reference's row order ----
      double F[3][FH_MAX_SEG], Qv[3][FH_MAX_SEG];
      bool acc[3] = {false, false, false};
      double rd[3] = {1.0, 1.0, 1.0}, ro[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int j = 0; j < 3; j++)
        for (int s = 0; s < N; s++) {
          const int kind = j + koff;
          F[j][s] = j < nrow ? basis_wcoef(kind == 0 ? 0 : (kind == 1 ? 3 : 4), N - 1 - s, 1.0) : 0.0;
          Qv[j][s] = 0.0;
        }
      for (int j = 0; j < nrow; j++) {
        double zj[FH_MAX_SEG], dd[3] = {0, 0, 0};
        for (int s = 0; s < N; s++) zj[s] = F[j][s];
        for (int pass = 0; pass < 2; pass++)
          for (int jp = 0; jp < j; jp++)
            if (acc[jp]) {
              double e = 0.0;
              for (int s = 0; s < N; s++) e += Qv[jp][s] * zj[s];
              for (int s = 0; s < N; s++) zj[s] -= e * Qv[jp][s];
              dd[jp] += e;
            }
        double zz = 0.0, f2 = 0.0;
        for (int s = 0; s < N; s++) { zz += zj[s] * zj[s]; f2 += F[j][s] * F[j][s]; }
        for (int jp = 0; jp < 3; jp++) ro[jp][j] = dd[jp];
        if (zz > dep2 * f2) {
          acc[j] = true;
          rd[j] = std::sqrt(zz);
          for (int s = 0; s < N; s++) Qv[j][s] = zj[s] / rd[j];
        }
      }
      // y = L rho (forward substitution), residual of a skipped row j: (G rho)_j
      double L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, G[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
      for (int j = 0; j < nrow; j++) {
        double row[3] = {0, 0, 0};  // rho_j - sum_jp ro[jp][j] y_jp as a combination of rho
        row[j] = 1.0;
        for (int jp = 0; jp < j; jp++)
          if (acc[jp])
            for (int c = 0; c < 3; c++) row[c] -= ro[jp][j] * L[jp][c];
        if (acc[j]) for (int c = 0; c < 3; c++) L[j][c] = row[c] / rd[j];
        else for (int c = 0; c < 3; c++) G[j][c] = row[c];
      }
      double* E = T + BT_EQ + ff * BT_EQ_WORDS;
      for (int l = 0; l < 3; l++)
        for (int c = 0; c < 3; c++) {
          double m = 0.0;  // M = B L, B[l][j] = Q[:, l] . Qv_j
          if (l < N)
            for (int j = 0; j < nrow; j++)
              if (acc[j]) {
                double b = 0.0;
                for (int s = 0; s < N; s++) b += Q[(size_t)s * N + l] * Qv[j][s];
                m += b * L[j][c];
              }
          E[l * 3 + c] = m;
          E[9 + l * 3 + c] = G[l][c];
        }
      double mask = 0.0;  // bit j: row j is linearly dependent on the rows before it (consistency check)
      for (int j = 0; j < nrow; j++)
        if (!acc[j]) mask += (double)(1 << j);
      E[18] = mask;
      // ---- inverse row norms in the reduced space ----
      const int c0 = 2 + ff, K = N - c0 > 0 ? N - c0 : 0;  // columns c0..N-1 of Q
      double* C = T + BT_C + ff * BT_C_WORDS;
      for (int kind = 0; kind < 5; kind++)
        for (int tt = 0; tt <= N; tt++) {
          double w2 = 0.0, o2 = 0.0;
          for (int s = 0; s < tt && s < N; s++) { const double w = basis_wcoef(kind, tt - 1 - s, 1.0); w2 += w * w; }
          for (int k = 0; k < K; k++) {
            double o = 0.0;
            for (int s = 0; s < tt && s < N; s++) o += Q[(size_t)s * N + c0 + k] * basis_wcoef(kind, tt - 1 - s, 1.0);
            o2 += o * o;
          }
          C[kind * BT_C_TT + tt] = (o2 > 1e-20 * w2 && o2 > 0.0) ? 1.0 / std::sqrt(o2) : 0.0;
        }
      double* CJ = T + BT_CJ + ff * BT_CJ_WORDS;
      for (int t = 0; t < N; t++) {
        double o2 = 0.0;
        for (int k = 0; k < K; k++) o2 += Q[(size_t)t * N + c0 + k] * Q[(size_t)t * N + c0 + k];
        CJ[t] = o2 > 1e-20 ? 1.0 / std::sqrt(o2) : 0.0;
      }
    }
  }
  return tab;
}

}  // namespace fh
