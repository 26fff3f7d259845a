/* faster_oracle.c — CPU restatement of the FASTER trajectory solver path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the arithmetic of the reference path lives in Gurobi Optimizer (closed source,
 * version not pinned: "8.1, 9.0, 9.1 tested", /root/reference/Readme.md:43), which is absent, and
 * the reference holds no test, golden vector or known-answer value for this path (SURVEY.md §4,
 * §8(c)).  This file restates the *model* that faster/src/solverGurobi.cpp hands to Gurobi and solves
 * it to global optimality with an exact method; it is pinned against an independent SciPy
 * implementation of the unreduced 12·N-coefficient model (oracle/py_model.py) and brute-force
 * enumeration of all P^N assignments (tests/test_oracle_*.py), not against Gurobi output.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 * The product (faster_amd/) never does.
 *
 * What is restated (file:line are relative to /root/reference/faster/):
 *   getDTInitial / findDT ............ src/solverGurobi.cpp:659-759, :494-497   -> orc_dt_initial
 *   MinPositiveElement ............... include/solverGurobi_utils.hpp:19-32     -> min_positive
 *   genNewTraj factor loop ........... src/solverGurobi.cpp:426-477             -> orc_solve
 *   model: createVars :70-84, setObjective :86-120, setConstraintsX0 :359-380,
 *          setConstraintsXf :332-357, setDynamicConstraints :499-524,
 *          setMaxConstraints :390-407, setPolytopesConstraints :180-291,
 *          control points getCP0..3 :833-862 ..............................      -> build_rows
 *   m.optimize() (callOptimizer :549-657) ....................................  -> miqp_bnb / gi_solve
 *   resetX / fillX .................... src/solverGurobi.cpp:382-388, :122-168  -> orc_sample
 *
 * Method.  The equalities (initial state, C2 continuity) are eliminated exactly by writing the
 * trajectory as a triple integrator driven by the per-segment jerk j_t = 6 a_t:
 * unknowns x = jerk in R^{3N}, objective sum (6a)^2 = |x|^2.  For a FIXED assignment the model is
 * a strictly convex QP  min |x|^2  s.t. E x = e, C x <= d, solved with a dual active-set method
 * (Goldfarb-Idnani 1983, specialised to an identity Hessian: the active normals are kept as a
 * thin QR, N = Q1 R, built by re-orthogonalised Gram-Schmidt, Givens rotations on removal).
 * The binaries b[t][p] with sum_p b[t][p] == 1 are resolved by branch and bound over
 * "segment t lies in polytope p": a node relaxes the indicator rows of unassigned segments; a node
 * whose optimum already has every unassigned segment inside some polytope is a leaf.  The bound is
 * exact (no MIP gap), so the result is the global optimum over all P^N assignments.
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/fasterhip.h"

#define NV_MAX (3 * FH_MAX_SEG)

typedef struct {
  double a[NV_MAX]; /* row of C: a.x <= rhs */
  double rhs;
  double nrm; /* |a| */
} orc_row;

typedef struct {
  int N, n;
  double h;
  const fh_problem* pr;
  const fh_face* faces; /* already offset by face_begin */
  fh_params par;
  /* coefficient of jerk j_s in the state at the START of segment t, m = t-1-s >= 0 */
  double cP[FH_MAX_SEG + 1], cV[FH_MAX_SEG + 1], cA;
  /* jerk-free part of the state at the start of segment t (t = 0..N) */
  double P0[FH_MAX_SEG + 1][3], V0[FH_MAX_SEG + 1][3], A0[FH_MAX_SEG + 1][3];
} orc_model;

/* ------------------------------------------------------------------------------------------------
 * Time allocation: getDTInitial (solverGurobi.cpp:659-759).  Intermediates are `float` exactly where
 * the reference declares them float (:662-670, :679-687, :718-720); the final division is float/int.
 * Roots: the reference uses Eigen::PolynomialSolver::realRoots (companion-matrix eigenvalues, a root
 * counts as real iff |imag| < 1e-12, Eigen's dummy_precision).  Eigen is absent; we use closed forms
 * polished by Newton steps, and the same |imag| threshold.
 * ---------------------------------------------------------------------------------------------- */
static double min_positive(const double* v, int n) { /* solverGurobi_utils.hpp:19-32 */
  double best = 0;
  int found = 0;
  for (int i = 0; i < n; i++)
    if (v[i] > 0 && (!found || v[i] < best)) {
      best = v[i];
      found = 1;
    }
  return found ? best : 0.0;
}

static double polish3(double c3, double c2, double c1, double c0, double t) {
  for (int it = 0; it < 3; it++) {
    double f = ((c3 * t + c2) * t + c1) * t + c0;
    double df = (3 * c3 * t + 2 * c2) * t + c1;
    if (df == 0 || !isfinite(df)) break;
    double tn = t - f / df;
    if (!isfinite(tn)) break;
    t = tn;
  }
  return t;
}

static int real_roots_quad(double c2, double c1, double c0, double* out);

/* real roots of c3 t^3 + c2 t^2 + c1 t + c0, c3 != 0; returns count */
static int real_roots_cubic(double c3, double c2, double c1, double c0, double* out) {
  /* [r6] CONVENTION (Eigen is absent, so its answer is not pinned): c0 == 0 exactly — start and goal coincide on this axis,
   * solverGurobi.cpp:691-693 builds the constant term x0 - xf — factors the cubic as t (c3 t^2 + c2 t + c1): one root is EXACTLY zero, which
   * MinPositiveElement (solverGurobi_utils.hpp:19-32: `v[i] > 0`) drops, and the others are the quadratic's roots by the quadratic's own
   * closed form.  The general closed form below returns that zero root as +1e-17, 0 or -1e-17 depending on the last bit of cbrt / acos
   * (two math libraries disagree on 0.025 % of such problems, and a companion-matrix eigenvalue solver would have a third opinion);
   * "the smallest positive root" must not hinge on that. */
  if (c0 == 0.0) {
    out[0] = 0.0;
    return 1 + real_roots_quad(c3, c2, c1, out + 1);
  }
  double B = c2 / c3, C = c1 / c3, D = c0 / c3;
  double p = C - B * B / 3.0;
  double q = 2.0 * B * B * B / 27.0 - B * C / 3.0 + D;
  double disc = q * q / 4.0 + p * p * p / 27.0;
  int k = 0;
  if (disc > 0) {
    double sq = sqrt(disc);
    double u = cbrt(-q / 2.0 + sq), v = cbrt(-q / 2.0 - sq);
    out[k++] = polish3(c3, c2, c1, c0, u + v - B / 3.0);
    double im = 0.5 * sqrt(3.0) * fabs(u - v);
    if (im < 1e-12) { /* Eigen would call the conjugate pair real */
      out[k++] = -(u + v) / 2.0 - B / 3.0;
      out[k++] = -(u + v) / 2.0 - B / 3.0;
    }
  } else if (p == 0) { /* disc<=0 and p==0 => q==0: triple root */
    out[k++] = -B / 3.0;
    out[k++] = -B / 3.0;
    out[k++] = -B / 3.0;
  } else {
    double m = 2.0 * sqrt(-p / 3.0);
    double arg = 3.0 * q / (p * m);
    if (arg > 1) arg = 1;
    if (arg < -1) arg = -1;
    double phi = acos(arg) / 3.0;
    for (int j = 0; j < 3; j++)
      out[k++] = polish3(c3, c2, c1, c0, m * cos(phi - 2.0 * M_PI * j / 3.0) - B / 3.0);
  }
  return k;
}

/* real roots of c2 t^2 + c1 t + c0, c2 != 0 */
static int real_roots_quad(double c2, double c1, double c0, double* out) {
  double disc = c1 * c1 - 4.0 * c2 * c0;
  if (disc < 0) {
    double im = sqrt(-disc) / fabs(2.0 * c2);
    if (im < 1e-12) {
      out[0] = out[1] = -c1 / (2.0 * c2);
      return 2;
    }
    return 0;
  }
  double s = sqrt(disc);
  double qq = -0.5 * (c1 + (c1 >= 0 ? s : -s));
  int k = 0;
  if (qq != 0) {
    out[k++] = qq / c2;
    out[k++] = c0 / qq;
  } else {
    out[k++] = 0;
    out[k++] = 0;
  }
  return k;
}

double orc_dt_initial(const fh_problem* pr) {
  const double* x0 = pr->x0;
  const double* xf = pr->xf;
  float tv[3], ta[3], tj[3];
  for (int i = 0; i < 3; i++) {
    tv[i] = (float)(fabs(xf[i] - x0[i]) / pr->v_max); /* :672-674 */
    float jerk = (float)(copysign(1.0, xf[i] - x0[i]) * pr->j_max); /* :679-681 */
    float a0 = (float)x0[6 + i], v0 = (float)x0[3 + i];             /* :682-687 */
    double r[3];
    int k = real_roots_cubic((double)jerk / 6.0, (double)a0 / 2.0, (double)v0, x0[i] - xf[i], r); /* :691-709 */
    tj[i] = (float)min_positive(r, k);                                                              /* :711-713 */
    float acc = (float)(copysign(1.0, xf[i] - x0[i]) * pr->a_max);                                  /* :718-720 */
    k = real_roots_quad(0.5 * (double)acc, (double)v0, x0[i] - xf[i], r);                           /* :724-742 */
    ta[i] = (float)min_positive(r, k);                                                              /* :744-746 */
  }
  float mx = tv[0];
  for (int i = 0; i < 3; i++) {
    if (tv[i] > mx) mx = tv[i];
    if (ta[i] > mx) mx = ta[i];
    if (tj[i] > mx) mx = tj[i];
  }
  double dt_initial = (double)(mx / (float)pr->n_seg); /* :751, float / int */
  if (dt_initial > 10000) dt_initial = 0;              /* :752-756 */
  return dt_initial;
}

/* ------------------------------------------------------------------------------------------------
 * Model in jerk space (equivalent to createVars + X0 + dynamic constraints).
 * ---------------------------------------------------------------------------------------------- */
static void model_init(orc_model* M, const fh_problem* pr, const fh_face* faces, const fh_params* par, double h) {
  M->N = pr->n_seg;
  M->n = 3 * pr->n_seg;
  M->h = h;
  M->pr = pr;
  M->faces = faces ? faces + pr->face_begin : NULL;
  M->par = *par;
  for (int m = 0; m <= M->N; m++) {
    M->cP[m] = h * h * h * (1.0 / 6.0 + 0.5 * m + 0.5 * m * (double)m);
    M->cV[m] = h * h * (0.5 + m);
  }
  M->cA = h;
  for (int i = 0; i < 3; i++) {
    M->P0[0][i] = pr->x0[i];
    M->V0[0][i] = pr->x0[3 + i];
    M->A0[0][i] = pr->x0[6 + i];
    for (int t = 0; t < M->N; t++) { /* zero-jerk propagation over one segment */
      M->P0[t + 1][i] = M->P0[t][i] + M->V0[t][i] * h + 0.5 * M->A0[t][i] * h * h;
      M->V0[t + 1][i] = M->V0[t][i] + M->A0[t][i] * h;
      M->A0[t + 1][i] = M->A0[t][i];
    }
  }
}

/* weights of (P_t, V_t, A_t, P_{t+1}) in Bezier control point k of segment t (getCP0..3, :833-862):
 * cp0 = P_t, cp1 = P_t + V_t h/3, cp2 = P_t + 2 V_t h/3 + A_t h^2/6, cp3 = P_{t+1}. */
static void cp_weights(double h, int k, double* wp, double* wv, double* wa, int* next) {
  *wp = 1;
  *wv = 0;
  *wa = 0;
  *next = 0;
  if (k == 1) *wv = h / 3.0;
  if (k == 2) {
    *wv = 2.0 * h / 3.0;
    *wa = h * h / 6.0;
  }
  if (k == 3) *next = 1;
}

/* row for: sum_i g[i] * (wp P + wv V + wa A)_{tt,i} <= bound, P/V/A at the start of segment tt */
static void make_row(const orc_model* M, int tt, const double g[3], double wp, double wv, double wa, double bound,
                     orc_row* r) {
  double c0 = 0;
  memset(r->a, 0, sizeof(r->a));
  for (int i = 0; i < 3; i++) {
    c0 += g[i] * (wp * M->P0[tt][i] + wv * M->V0[tt][i] + wa * M->A0[tt][i]);
    for (int s = 0; s < tt; s++) {
      int m = tt - 1 - s;
      r->a[3 * s + i] = g[i] * (wp * M->cP[m] + wv * M->cV[m] + wa * M->cA);
    }
  }
  r->rhs = bound - c0;
  double nn = 0;
  for (int v = 0; v < M->n; v++) nn += r->a[v] * r->a[v];
  r->nrm = sqrt(nn);
}

/* Build equality rows and inequality rows for a node (assign[t] = polytope or -1 = relaxed).
 * Returns 1 if a constant (jerk-independent) row is violated => node infeasible. */
static int build_rows(const orc_model* M, const int8_t* assign, orc_row* eq, int* me, orc_row* in, int* mi) {
  const fh_problem* pr = M->pr;
  const int N = M->N;
  const double tol = M->par.feas_tol;
  int ne = 0, ni = 0, bad = 0;
  /* final state, setConstraintsXf :332-357 (order per axis: [pos], vel, accel) */
  for (int i = 0; i < 3; i++) {
    double g[3] = {0, 0, 0};
    g[i] = 1;
    if (pr->force_final_pos) make_row(M, N, g, 1, 0, 0, pr->xf[i], &eq[ne++]);
    make_row(M, N, g, 0, 1, 0, pr->xf[3 + i], &eq[ne++]);
    make_row(M, N, g, 0, 0, 1, pr->xf[6 + i], &eq[ne++]);
  }
  /* setMaxConstraints :390-407: |vel(t,0)|<=v_max, |accel(t,0)|<=a_max, |jerk(t)|<=j_max */
  for (int t = 0; t < N; t++)
    for (int i = 0; i < 3; i++)
      for (int sg = -1; sg <= 1; sg += 2) {
        double g[3] = {0, 0, 0};
        g[i] = sg;
        orc_row r;
        make_row(M, t, g, 0, 1, 0, pr->v_max, &r);
        if (r.nrm == 0) bad |= (-r.rhs > tol); else in[ni++] = r;
        make_row(M, t, g, 0, 0, 1, pr->a_max, &r);
        if (r.nrm == 0) bad |= (-r.rhs > tol); else in[ni++] = r;
        memset(&r, 0, sizeof(r));
        r.a[3 * t + i] = sg;
        r.rhs = pr->j_max;
        r.nrm = 1;
        in[ni++] = r;
      }
  /* setPolytopesConstraints :237-289 with the binaries of assigned segments fixed */
  for (int t = 0; t < N; t++) {
    if (assign[t] < 0) continue;
    int p = assign[t];
    for (int f = pr->face_off[p]; f < pr->face_off[p + 1]; f++)
      for (int k = 0; k < 4; k++) {
        double wp, wv, wa;
        int nx;
        cp_weights(M->h, k, &wp, &wv, &wa, &nx);
        orc_row r;
        make_row(M, t + nx, M->faces[f].a, wp, wv, wa, M->faces[f].b, &r);
        if (r.nrm == 0) bad |= (-r.rhs > tol); else in[ni++] = r;
      }
  }
  *me = ne;
  *mi = ni;
  return bad;
}

/* ------------------------------------------------------------------------------------------------
 * Dual active-set QP:  min |x|^2  s.t.  eq rows a.x = rhs, in rows a.x <= rhs.
 * returns 0 optimal, 1 infeasible, 2 lower bound reached `ub` (pruned), 3 iteration limit
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int n, q;
  double x[NV_MAX];
  double Q[NV_MAX][NV_MAX]; /* Q[i][c]: column c of Q1 */
  double R[NV_MAX][NV_MAX];
  double u[NV_MAX];
  int act[NV_MAX]; /* >=0: inequality index; <0: equality -(e+1) */
} gi_state;

/* z = (I - Q1 Q1^T) g (CGS2), d = Q1^T g; returns |z|^2 */
static double gi_project(const gi_state* S, const double* g, double* z, double* d) {
  int n = S->n, q = S->q;
  for (int c = 0; c < q; c++) {
    double s = 0;
    for (int i = 0; i < n; i++) s += S->Q[i][c] * g[i];
    d[c] = s;
  }
  for (int i = 0; i < n; i++) {
    double s = g[i];
    for (int c = 0; c < q; c++) s -= S->Q[i][c] * d[c];
    z[i] = s;
  }
  for (int c = 0; c < q; c++) { /* second pass */
    double s = 0;
    for (int i = 0; i < n; i++) s += S->Q[i][c] * z[i];
    d[c] += s;
    for (int i = 0; i < n; i++) z[i] -= S->Q[i][c] * s;
  }
  double zz = 0;
  for (int i = 0; i < n; i++) zz += z[i] * z[i];
  return zz;
}

static void gi_backsolve(const gi_state* S, const double* d, double* r) {
  for (int c = S->q - 1; c >= 0; c--) {
    double s = d[c];
    for (int k = c + 1; k < S->q; k++) s -= S->R[c][k] * r[k];
    r[c] = s / S->R[c][c];
  }
}

static void gi_add(gi_state* S, const double* z, double zz, const double* d, int id, double u) {
  int q = S->q;
  double rho = sqrt(zz);
  for (int i = 0; i < S->n; i++) S->Q[i][q] = z[i] / rho;
  for (int c = 0; c < q; c++) S->R[c][q] = d[c];
  for (int c = 0; c <= q; c++) S->R[q][c] = 0;
  S->R[q][q] = rho;
  S->act[q] = id;
  S->u[q] = u;
  S->q = q + 1;
}

static void gi_drop(gi_state* S, int k) {
  int q = S->q, n = S->n;
  for (int c = k; c < q - 1; c++) {
    S->act[c] = S->act[c + 1];
    S->u[c] = S->u[c + 1];
    for (int rr = 0; rr < q; rr++) S->R[rr][c] = S->R[rr][c + 1];
  }
  for (int j = k; j < q - 1; j++) { /* zero R[j+1][j] */
    double a = S->R[j][j], b = S->R[j + 1][j];
    double rr = hypot(a, b);
    if (rr == 0) continue;
    double cs = a / rr, sn = b / rr;
    for (int c = j; c < q - 1; c++) {
      double t1 = S->R[j][c], t2 = S->R[j + 1][c];
      S->R[j][c] = cs * t1 + sn * t2;
      S->R[j + 1][c] = -sn * t1 + cs * t2;
    }
    for (int i = 0; i < n; i++) {
      double t1 = S->Q[i][j], t2 = S->Q[i][j + 1];
      S->Q[i][j] = cs * t1 + sn * t2;
      S->Q[i][j + 1] = -sn * t1 + cs * t2;
    }
  }
  S->q = q - 1;
}

static int gi_solve(int n, int me, const orc_row* eq, int mi, const orc_row* in, const fh_params* par, double ub,
                    double* xout, double* cost, int* iters) {
  static __thread gi_state Sst;
  gi_state* S = &Sst;
  S->n = n;
  S->q = 0;
  memset(S->x, 0, sizeof(S->x));
  const double tol = par->feas_tol, dep2 = par->dep_tol * par->dep_tol;
  double z[NV_MAX], d[NV_MAX], r[NV_MAX];
  /* per-thread scratch that only grows: large per-solve malloc/free pairs turn into mmap/munmap and do not scale */
  static __thread char* active_buf = NULL;
  static __thread size_t active_cap = 0;
  if ((size_t)mi + 1 > active_cap) {
    free(active_buf);
    active_cap = 2 * ((size_t)mi + 1);
    active_buf = (char*)malloc(active_cap);
  }
  char* active = active_buf;
  memset(active, 0, (size_t)mi + 1);
  int it = 0, status = -1;

  /* equalities first: always a full step, never dropped */
  for (int e = 0; e < me && status < 0; e++) {
    double v = -eq[e].rhs;
    for (int i = 0; i < n; i++) v += eq[e].a[i] * S->x[i];
    double zz = gi_project(S, eq[e].a, z, d);
    if (zz <= dep2 * eq[e].nrm * eq[e].nrm) {
      if (fabs(v) > tol) status = 1;
      continue;
    }
    double t = v / zz;
    gi_backsolve(S, d, r);
    for (int i = 0; i < n; i++) S->x[i] -= t * z[i];
    for (int c = 0; c < S->q; c++) S->u[c] -= t * r[c];
    gi_add(S, z, zz, d, -(e + 1), t);
    it++;
  }

  while (status < 0) {
    double f = 0;
    for (int i = 0; i < n; i++) f += S->x[i] * S->x[i];
    if (f >= ub) {
      status = 2;
      break;
    }
    /* most violated inactive row, violation measured relative to the row norm */
    int p = -1;
    double best = 0, vp = 0;
    for (int r_ = 0; r_ < mi; r_++) {
      if (active[r_]) continue;
      double v = -in[r_].rhs;
      for (int i = 0; i < n; i++) v += in[r_].a[i] * S->x[i];
      if (v > tol && v / in[r_].nrm > best) {
        best = v / in[r_].nrm;
        p = r_;
        vp = v;
      }
    }
    if (p < 0) {
      status = 0;
      break;
    }
    double up = 0;
    for (;;) { /* until row p is active (or infeasible) */
      if (++it > par->max_iters) {
        status = 3;
        break;
      }
      double zz = gi_project(S, in[p].a, z, d);
      gi_backsolve(S, d, r);
      int dependent = zz <= dep2 * in[p].nrm * in[p].nrm;
      /* dual blocking ratio */
      int kb = -1;
      double t1 = INFINITY;
      for (int c = 0; c < S->q; c++)
        if (S->act[c] >= 0 && r[c] > 0) {
          double tt = S->u[c] / r[c];
          if (tt < t1) {
            t1 = tt;
            kb = c;
          }
        }
      double t2 = dependent ? INFINITY : vp / zz;
      if (kb < 0 && dependent) {
        status = 1; /* infeasible */
        break;
      }
      double t = t1 < t2 ? t1 : t2;
      for (int c = 0; c < S->q; c++) S->u[c] -= t * r[c];
      up += t;
      if (!dependent) {
        for (int i = 0; i < n; i++) S->x[i] -= t * z[i];
        vp -= t * zz;
      }
      if (t2 <= t1) { /* full step */
        gi_add(S, z, zz, d, p, up);
        active[p] = 1;
        break;
      }
      active[S->act[kb]] = 0;
      gi_drop(S, kb);
    }
  }
  double f = 0;
  for (int i = 0; i < n; i++) {
    xout[i] = S->x[i];
    f += S->x[i] * S->x[i];
  }
  *cost = f;
  *iters += it;
  return status;
}

/* ------------------------------------------------------------------------------------------------
 * States / control points from a jerk vector
 * ---------------------------------------------------------------------------------------------- */
static void states_from_x(const orc_model* M, const double* x, double P[][3], double V[][3], double A[][3]) {
  double h = M->h;
  for (int i = 0; i < 3; i++) {
    P[0][i] = M->pr->x0[i];
    V[0][i] = M->pr->x0[3 + i];
    A[0][i] = M->pr->x0[6 + i];
    for (int t = 0; t < M->N; t++) {
      double j = x[3 * t + i];
      P[t + 1][i] = P[t][i] + V[t][i] * h + 0.5 * A[t][i] * h * h + j * h * h * h / 6.0;
      V[t + 1][i] = V[t][i] + A[t][i] * h + 0.5 * j * h * h;
      A[t + 1][i] = A[t][i] + j * h;
    }
  }
}

/* max over faces f of polytope p and control points k of (A_f . cp_k(t) - b_f) */
static double seg_poly_violation(const orc_model* M, double P[][3], double V[][3], double A[][3], int t, int p) {
  const fh_problem* pr = M->pr;
  double h = M->h, worst = -INFINITY;
  for (int k = 0; k < 4; k++) {
    double wp, wv, wa, cp[3];
    int nx;
    cp_weights(h, k, &wp, &wv, &wa, &nx);
    for (int i = 0; i < 3; i++) cp[i] = wp * P[t + nx][i] + wv * V[t + nx][i] + wa * A[t + nx][i];
    for (int f = pr->face_off[p]; f < pr->face_off[p + 1]; f++) {
      const fh_face* F = &M->faces[f];
      double v = F->a[0] * cp[0] + F->a[1] * cp[1] + F->a[2] * cp[2] - F->b;
      if (v > worst) worst = v;
    }
  }
  return worst;
}

/* ------------------------------------------------------------------------------------------------
 * MIQP for one dt: branch and bound over the segment->polytope assignment (replaces m.optimize()).
 * returns FH_ST_OPTIMAL / FH_ST_INFEASIBLE / FH_ST_NODE_LIMIT / FH_ST_ITER_LIMIT
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  orc_model M;
  orc_row *eq, *in;
  double best_cost;
  double best_x[NV_MAX];
  int8_t best_assign[FH_MAX_SEG];
  int nodes, iters, limit, iters_before;
  int early; /* branching rule of this trial, decided at its root (bnb_node) */
  unsigned allowed[FH_MAX_SEG]; /* bit p set: polytope p is not excluded for segment t by jerk-independent rows */
} bnb_ctx;

/* Exact screening on jerk-independent indicator rows.  Control points 0..2 of segment 0 are functions of x0
 * and h only; with the final position forced, control points 1..3 of the last segment are functions of xf
 * and h only (cp3 = pf, cp2 = pf - vf h/3, cp1 = pf - 2 vf h/3 + af h^2/6).  A polytope that does not hold them
 * can never be chosen for that segment (b[t][p] = 1 would violate its indicator rows, :283-286). */
static void screen_constant_rows(bnb_ctx* B) {
  const orc_model* M = &B->M;
  const fh_problem* pr = M->pr;
  const double h = M->h, tol = M->par.feas_tol;
  const int N = M->N;
  for (int t = 0; t < N; t++) B->allowed[t] = pr->n_poly ? ((1u << pr->n_poly) - 1u) : 0u;
  if (!pr->n_poly) return;
  double cp[2][3][3];
  for (int i = 0; i < 3; i++) {
    const double p0 = pr->x0[i], v0 = pr->x0[3 + i], a0 = pr->x0[6 + i];
    cp[0][0][i] = p0;
    cp[0][1][i] = p0 + v0 * (h / 3.0);
    cp[0][2][i] = p0 + v0 * (2.0 * h / 3.0) + a0 * (h * h / 6.0);
    const double pf = pr->xf[i], vf = pr->xf[3 + i], af = pr->xf[6 + i];
    cp[1][0][i] = pf;
    cp[1][1][i] = pf - vf * (h / 3.0);
    cp[1][2][i] = pf - vf * (2.0 * h / 3.0) + af * (h * h / 6.0);
  }
  for (int e = 0; e < (pr->force_final_pos ? 2 : 1); e++) {
    const int t = e ? N - 1 : 0;
    for (int p = 0; p < pr->n_poly; p++) {
      double worst = -INFINITY;
      for (int f = pr->face_off[p]; f < pr->face_off[p + 1]; f++)
        for (int k = 0; k < 3; k++) {
          const fh_face* F = &M->faces[f];
          const double v = F->a[0] * cp[e][k][0] + F->a[1] * cp[e][k][1] + F->a[2] * cp[e][k][2] - F->b;
          if (v > worst) worst = v;
        }
      if (worst > tol) B->allowed[t] &= ~(1u << p);
    }
  }
}

static void bnb_node(bnb_ctx* B, int8_t* assign) {
  const orc_model* M = &B->M;
  const fh_problem* pr = M->pr;
  if (B->limit) return;
  if (B->nodes >= M->par.max_nodes) {
    B->limit = FH_ST_NODE_LIMIT;
    return;
  }
  if (M->par.max_work > 0 && B->iters_before + B->iters >= M->par.max_work) {
    B->limit = FH_ST_ITER_LIMIT;
    return;
  }
  B->nodes++;
  int me, mi;
  if (build_rows(M, assign, B->eq, &me, B->in, &mi)) return;
  double x[NV_MAX], cost;
  /* mip_gap = 0 (default): exact.  > 0: Gurobi's MIPGap rule, a node within the relative gap of the incumbent is pruned */
  int st = gi_solve(M->n, me, B->eq, mi, B->in, &M->par, B->best_cost * (1.0 - M->par.mip_gap), x, &cost, &B->iters);
  if (st == 3) {
    B->limit = FH_ST_ITER_LIMIT;
    return;
  }
  if (st != 0) return; /* infeasible or bounded out */
  /* Which unassigned segment to branch on: the one that is least inside any polytope (the quickest way to a good leaf) — unless the
   * ROOT relaxation of this trial ends outside the corridor (its last segment is inside no polytope: a trajectory that cannot stop
   * in time, the typical infeasible safe problem).  Below the root of such a trial the EARLIEST violated segment is taken: the
   * trajectory is causal (segment t depends on the jerks 0..t only), so deciding the early segments first makes the children's QPs
   * tight and an infeasible trial is refuted in a fraction of the nodes; when the root overshoots the corridor by more than 1.2
   * braking distances from v_max (v_max^2 / 2 a_max), the root itself branches that way too.  Config C5's safe problems that no
   * factor solves: 621 -> 19 nodes (11663 -> 265 active-set iterations); trials whose root ends inside the corridor — every whole
   * problem — keep their trees node for node.  The rule is a function of the trial's root alone, not of the order in which the
   * tree is explored.  Any rule is exact: it only orders the search. */
  const int root = B->nodes == 1;
  double P[FH_MAX_SEG + 1][3], V[FH_MAX_SEG + 1][3], A[FH_MAX_SEG + 1][3];
  states_from_x(M, x, P, V, A);
  int8_t full[FH_MAX_SEG];
  int bseg = -1, fseg = -1; /* the segment least inside any polytope; the earliest violated one */
  double bworst = M->par.feas_tol;
  double viol[FH_MAX_POLY];
  double bviol[FH_MAX_POLY], fviol[FH_MAX_POLY];
  int root_early = 0;
  for (int t = 0; t < M->N; t++) {
    full[t] = assign[t];
    if (assign[t] >= 0 || pr->n_poly == 0) continue;
    double mn = INFINITY;
    int arg = 0;
    for (int p = 0; p < pr->n_poly; p++) {
      viol[p] = ((B->allowed[t] >> p) & 1u) ? seg_poly_violation(M, P, V, A, t, p) : INFINITY;
      if (viol[p] < mn) {
        mn = viol[p];
        arg = p;
      }
    }
    full[t] = (int8_t)arg;
    if (root && t == M->N - 1) {
      B->early = mn > M->par.feas_tol;
      root_early = mn > 1.2 * (pr->v_max * pr->v_max) / (2.0 * pr->a_max);
    }
    if (fseg < 0 && mn > M->par.feas_tol) {
      fseg = t;
      memcpy(fviol, viol, sizeof(viol));
    }
    if (mn > bworst) {
      bworst = mn;
      bseg = t;
      memcpy(bviol, viol, sizeof(viol));
    }
  }
  if (root ? root_early : B->early) {
    bseg = fseg;
    memcpy(bviol, fviol, sizeof(fviol));
  }
  if (bseg < 0) { /* leaf: feasible for the MIQP */
    if (cost < B->best_cost) {
      B->best_cost = cost;
      memcpy(B->best_x, x, sizeof(x));
      memcpy(B->best_assign, full, sizeof(full));
    }
    return;
  }
  /* branch on segment bseg, most promising polytope first */
  int order[FH_MAX_POLY];
  for (int p = 0; p < pr->n_poly; p++) order[p] = p;
  for (int a = 1; a < pr->n_poly; a++) /* insertion sort, stable */
    for (int b = a; b > 0 && bviol[order[b]] < bviol[order[b - 1]]; b--) {
      int tmp = order[b];
      order[b] = order[b - 1];
      order[b - 1] = tmp;
    }
#ifdef ORC_PARENT_BOUND
  /* Experimental (make CFLAGS+=-DORC_PARENT_BOUND; the kernels' -DFH_PARENT_BOUND): a child is not visited when a lower bound of
   * its QP, known at this node, already loses against the incumbent.  The multipliers of this node's optimum x* together with one
   * multiplier on a single row n of the child (violation v > 0 at x*) are dual feasible for the child's QP; the best such
   * multiplier gives cost* + v^2 / |n'|^2, n' = n projected off the equality rows (their multipliers are free).  v is taken
   * beyond the feasibility tolerance, as the kernels' normalised rows carry it. */
  double qeq[9][NV_MAX];
  int nq = 0;
  for (int e = 0; e < me; e++) { /* orthonormal basis of the equality rows (Gram-Schmidt, twice) */
    double v[NV_MAX];
    memcpy(v, B->eq[e].a, sizeof(double) * M->n);
    for (int pass = 0; pass < 2; pass++)
      for (int j = 0; j < nq; j++) {
        double d = 0;
        for (int i = 0; i < M->n; i++) d += qeq[j][i] * v[i];
        for (int i = 0; i < M->n; i++) v[i] -= d * qeq[j][i];
      }
    double nn = 0;
    for (int i = 0; i < M->n; i++) nn += v[i] * v[i];
    if (nn > 1e-20) {
      nn = sqrt(nn);
      for (int i = 0; i < M->n; i++) qeq[nq][i] = v[i] / nn;
      nq++;
    }
  }
#endif
  for (int c = 0; c < pr->n_poly; c++) {
    if (!((B->allowed[bseg] >> order[c]) & 1u)) continue;
#ifdef ORC_PARENT_BOUND
    if (B->best_cost < INFINITY) {
      const int p = order[c];
      double best = 0;
      for (int f = pr->face_off[p]; f < pr->face_off[p + 1]; f++) {
        const double* a3 = M->faces[f].a;
        const double na3 = sqrt(a3[0] * a3[0] + a3[1] * a3[1] + a3[2] * a3[2]);
        for (int k = 0; k < 4; k++) {
          double wp, wv, wa;
          int nx;
          cp_weights(M->h, k, &wp, &wv, &wa, &nx);
          orc_row r;
          make_row(M, bseg + nx, a3, wp, wv, wa, M->faces[f].b, &r);
          if (r.nrm == 0) continue;
          double ax = 0;
          for (int v = 0; v < M->n; v++) ax += r.a[v] * x[v];
          const double vio = ax - r.rhs - M->par.feas_tol * (na3 > 0 ? 1.0 : 0.0);
          if (!(vio > 0)) continue;
          double n2 = r.nrm * r.nrm;
          for (int j = 0; j < nq; j++) {
            double d = 0;
            for (int i = 0; i < M->n; i++) d += qeq[j][i] * r.a[i];
            n2 -= d * d;
          }
          if (n2 < 1e-12 * r.nrm * r.nrm) continue; /* a row that does not depend on the free unknowns */
          const double bnd = vio * vio / n2;
          if (bnd > best) best = bnd;
        }
      }
      if (cost + best * (1.0 - 1e-9) >= B->best_cost * (1.0 - M->par.mip_gap)) continue;
    }
#endif
    assign[bseg] = (int8_t)order[c];
    bnb_node(B, assign);
  }
  assign[bseg] = -1;
}

static int miqp_bnb(const fh_problem* pr, const fh_face* faces, const fh_params* par, double h, double* x,
                    double* cost, int8_t* assign_out, int* nodes, int* iters, const int8_t* fixed_assign) {
  bnb_ctx B;
  model_init(&B.M, pr, faces, par, h);
  int nf = 0;
  for (int p = 0; p < pr->n_poly; p++) {
    int c = pr->face_off[p + 1] - pr->face_off[p];
    if (c > nf) nf = c;
  }
  size_t max_in = (size_t)18 * pr->n_seg + (size_t)4 * pr->n_seg * (nf > 0 ? nf : 1) + 8;
  static __thread orc_row* eq_buf = NULL;
  static __thread orc_row* in_buf = NULL;
  static __thread size_t in_cap = 0;
  if (!eq_buf) eq_buf = (orc_row*)malloc(sizeof(orc_row) * 9);
  if (max_in > in_cap) {
    free(in_buf);
    in_cap = 2 * max_in;
    in_buf = (orc_row*)malloc(sizeof(orc_row) * in_cap);
  }
  B.eq = eq_buf;
  B.in = in_buf;
  B.best_cost = INFINITY;
  B.nodes = 0;
  B.early = 0;
  B.iters = 0;
  B.iters_before = *iters;
  B.limit = 0;
  int8_t assign[FH_MAX_SEG];
  for (int t = 0; t < FH_MAX_SEG; t++) assign[t] = fixed_assign ? fixed_assign[t] : -1;
  screen_constant_rows(&B);
  int screened_out = 0;
  for (int t = 0; t < pr->n_seg && pr->n_poly; t++) {
    if (assign[t] >= 0) {
      if (assign[t] >= pr->n_poly || !((B.allowed[t] >> assign[t]) & 1u)) screened_out = 1;
    } else if (B.allowed[t] == 0u) {
      screened_out = 1;
    } else if ((B.allowed[t] & (B.allowed[t] - 1u)) == 0u) { /* exactly one candidate: not a decision */
      int p = 0;
      while (!((B.allowed[t] >> p) & 1u)) p++;
      assign[t] = (int8_t)p;
    }
  }
  if (!screened_out) bnb_node(&B, assign);
  *nodes += B.nodes;
  *iters += B.iters;
  if (B.limit) return B.limit;
  if (!(B.best_cost < INFINITY)) return FH_ST_INFEASIBLE;
  memcpy(x, B.best_x, sizeof(double) * 3 * pr->n_seg);
  *cost = B.best_cost;
  for (int t = 0; t < pr->n_seg; t++) assign_out[t] = pr->n_poly ? B.best_assign[t] : -1;
  return FH_ST_OPTIMAL;
}

/* polynomial coefficients in the reference variable order (createVars :70-84) */
static void coeff_from_x(const fh_problem* pr, double h, const double* x, double coeff[][12]) {
  orc_model M;
  fh_params dummy = {0};
  model_init(&M, pr, NULL, &dummy, h);
  double P[FH_MAX_SEG + 1][3], V[FH_MAX_SEG + 1][3], A[FH_MAX_SEG + 1][3];
  states_from_x(&M, x, P, V, A);
  for (int t = 0; t < pr->n_seg; t++)
    for (int i = 0; i < 3; i++) {
      coeff[t][0 + i] = x[3 * t + i] / 6.0;
      coeff[t][3 + i] = A[t][i] / 2.0;
      coeff[t][6 + i] = V[t][i];
      coeff[t][9 + i] = P[t][i];
    }
}

static int bad_input(const fh_problem* pr) {
  if (pr->n_seg < 1 || pr->n_seg > FH_MAX_SEG || pr->n_poly < 0 || pr->n_poly > FH_MAX_POLY) return 1;
  if (pr->face_off[0] != 0 || pr->face_begin < 0) return 1;
  for (int p = 0; p < pr->n_poly; p++) {
    int c = pr->face_off[p + 1] - pr->face_off[p];
    if (c < 0 || c > FH_MAX_FACES_POLY) return 1;
  }
  if (pr->n_poly && pr->face_off[pr->n_poly] > FH_MAX_FACES) return 1;
  if (!(pr->f_inc > 0) || !isfinite(pr->f_init) || !isfinite(pr->f_final)) return 1;
  if ((pr->f_final - pr->f_init) / pr->f_inc > 4096.0) return 1; /* FH_MAX_TRIALS */
  if (!(pr->dc > 0) || !(pr->v_max > 0) || !(pr->a_max > 0) || !(pr->j_max > 0)) return 1;
  for (int i = 0; i < 9; i++)
    if (!isfinite(pr->x0[i]) || !isfinite(pr->xf[i])) return 1;
  {
    const unsigned long long pins = (unsigned long long)pr->pin[0] | ((unsigned long long)pr->pin[1] << 32);
    for (int t = 0; t < FH_MAX_SEG; t++) {
      const int v = (int)((pins >> (4 * t)) & 15ull);
      if (v && (t >= pr->n_seg || v > pr->n_poly)) return 1;
    }
  }
  return 0;
}

/* genNewTraj (solverGurobi.cpp:426-477).  If fixed_assign != NULL the binaries are fixed (BASELINE
 * config 1 "fixed binaries (pure QP)"): entries >= 0 pin a segment, -1 leaves it free. */
void orc_solve_fixed(const fh_problem* pr, const fh_face* faces, const fh_params* par, const int8_t* fixed_assign,
                     fh_result* res) {
  memset(res, 0, sizeof(*res));
  for (int t = 0; t < FH_MAX_SEG; t++) res->assign[t] = -1;
  if (bad_input(pr)) {
    res->status = FH_ST_BAD_INPUT;
    return;
  }
  double dt_init = orc_dt_initial(pr);
  int status = FH_ST_INFEASIBLE;
  double x[NV_MAX];
  for (double f = pr->f_init; f <= pr->f_final && !res->solved && !(par->max_work > 0 && status == FH_ST_ITER_LIMIT);
       f = f + pr->f_inc) { /* :445-446 */
    res->trials++;
    double two_dc = 2 * pr->dc;
    res->dt = f * (dt_init > two_dc ? dt_init : two_dc); /* findDT :494-497 */
    status = miqp_bnb(pr, faces, par, res->dt, x, &res->cost, res->assign, &res->nodes, &res->qp_iters, fixed_assign);
    if (status == FH_ST_OPTIMAL) {
      res->solved = 1;
      res->factor = f;
      coeff_from_x(pr, res->dt, x, res->coeff);
    }
  }
  res->status = status;
  if (!res->solved) res->cost = 0;
}

void orc_solve(const fh_problem* pr, const fh_face* faces, const fh_params* par, fh_result* res) {
  const unsigned long long pins = (unsigned long long)pr->pin[0] | ((unsigned long long)pr->pin[1] << 32);
  if (!pins) {
    orc_solve_fixed(pr, faces, par, NULL, res);
    return;
  }
  int8_t fixed[FH_MAX_SEG];
  for (int t = 0; t < FH_MAX_SEG; t++) fixed[t] = (int8_t)((int)((pins >> (4 * t)) & 15ull) - 1);
  orc_solve_fixed(pr, faces, par, fixed, res);
}

/* threads <= 0: the OpenMP default (all cores) */
void orc_solve_batch_mt(const fh_problem* pr, const fh_face* faces, const fh_params* par, int n, fh_result* res, int threads) {
  if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
  for (int i = 0; i < n; i++) orc_solve(&pr[i], faces, par, &res[i]);
}

void orc_solve_batch(const fh_problem* pr, const fh_face* faces, const fh_params* par, int n, fh_result* res) {
  orc_solve_batch_mt(pr, faces, par, n, res, 0);
}

int orc_max_threads(void) { return omp_get_max_threads(); }

/* One MIQP for a given dt (one callOptimizer()), optionally with pinned segments. Returns FH_ST_*. */
int orc_miqp_dt(const fh_problem* pr, const fh_face* faces, const fh_params* par, double dt,
                const int8_t* fixed_assign, fh_result* res) {
  memset(res, 0, sizeof(*res));
  double x[NV_MAX];
  res->dt = dt;
  res->trials = 1;
  int st = miqp_bnb(pr, faces, par, dt, x, &res->cost, res->assign, &res->nodes, &res->qp_iters, fixed_assign);
  res->status = st;
  if (st == FH_ST_OPTIMAL) {
    res->solved = 1;
    coeff_from_x(pr, dt, x, res->coeff);
  }
  return st;
}

/* resetX + fillX (solverGurobi.cpp:382-388, :122-168). Returns the sample count; writes at most
 * max_samples states. */
int orc_sample(const fh_problem* pr, const fh_result* res, int max_samples, fh_state* out) {
  if (!res->solved) return 0;
  int N = pr->n_seg;
  double dt = res->dt, DC = pr->dc;
  int size = (int)((int)(N)*dt / DC); /* :384 */
  size = (size < 2) ? 2 : size;       /* :385 */
  double t = 0;
  int interval = 0;
  for (int i = 0; i < size; i++) {
    t = t + DC;                                                        /* :131 */
    if (t > dt * (interval + 1)) interval = (interval + 1 < N - 1) ? interval + 1 : N - 1; /* :132-135 */
    if (i >= max_samples) continue;
    double tau = t - interval * dt;
    const double* c = res->coeff[interval];
    fh_state s;
    for (int a = 0; a < 3; a++) {
      s.pos[a] = c[0 + a] * tau * tau * tau + c[3 + a] * tau * tau + c[6 + a] * tau + c[9 + a]; /* getPos :761-767 */
      s.vel[a] = 3 * c[0 + a] * tau * tau + 2 * c[3 + a] * tau + c[6 + a];                     /* getVel :769-774 */
      s.accel[a] = 6 * c[0 + a] * tau + 2 * c[3 + a];                                          /* getAccel :776-781 */
      s.jerk[a] = 6 * c[0 + a];                                                                /* getJerk :783-788 */
    }
    if (i == size - 1) /* :165-167 */
      for (int a = 0; a < 3; a++) s.vel[a] = s.accel[a] = s.jerk[a] = 0;
    out[i] = s;
  }
  return size;
}

/* Brute force over ALL P^N assignments (validation of the branch and bound on small cases).
 * Returns number of feasible assignments; best cost/assignment in res. */
long orc_bruteforce_dt(const fh_problem* pr, const fh_face* faces, const fh_params* par, double dt, fh_result* res) {
  memset(res, 0, sizeof(*res));
  int N = pr->n_seg, P = pr->n_poly;
  long total = 1, feasible = 0;
  for (int t = 0; t < N; t++) total *= P;
  double best = INFINITY;
  res->dt = dt;
  for (long id = 0; id < total; id++) {
    int8_t a[FH_MAX_SEG];
    long r = id;
    for (int t = N - 1; t >= 0; t--) {
      a[t] = (int8_t)(r % P);
      r /= P;
    }
    fh_result one;
    fh_params p2 = *par;
    int st = orc_miqp_dt(pr, faces, &p2, dt, a, &one);
    res->nodes += one.nodes;
    if (st == FH_ST_OPTIMAL) {
      feasible++;
      if (one.cost < best) {
        best = one.cost;
        int nn = res->nodes;
        *res = one;
        res->nodes = nn;
      }
    }
  }
  if (!(best < INFINITY)) res->status = FH_ST_INFEASIBLE;
  return feasible;
}

/* The Bezier control points of a result, by the ORACLE's route (the jerk-space form it builds its rows from, cp_weights above):
 * with P = d, V = c, A = 2b, j = 6a at the start of a segment of length h = dt,
 *   cp0 = P, cp1 = P + V h/3, cp2 = P + 2 V h/3 + A h^2/6, cp3 = P + V h + A h^2/2 + j h^3/6
 * — algebraically getCP0..3 (solverGurobi.cpp:833-862), evaluated otherwise than the reference's literal expressions, which the
 * product's fh_control_points follows.  cp: [n_seg][4][3]; zero if the result is unsolved. */
static void control_points_one(const fh_result* res, int n_seg, double* cp) {
  memset(cp, 0, sizeof(double) * (size_t)n_seg * 12);
  if (!res->solved) return;
  const double h = res->dt;
  for (int t = 0; t < n_seg; t++)
    for (int i = 0; i < 3; i++) {
      const double P = res->coeff[t][9 + i], V = res->coeff[t][6 + i], A = 2.0 * res->coeff[t][3 + i], j = 6.0 * res->coeff[t][0 + i];
      double* o = cp + (size_t)t * 12;
      for (int k = 0; k < 3; k++) {
        double wp, wv, wa;
        int next;
        cp_weights(h, k, &wp, &wv, &wa, &next);
        o[3 * k + i] = wp * P + wv * V + wa * A;
      }
      o[9 + i] = P + V * h + 0.5 * A * h * h + j * h * h * h / 6.0;
    }
}

void orc_control_points(const fh_result* res, int n, int n_seg, double* cp) {
  for (int i = 0; i < n; i++) control_points_one(res + i, n_seg, cp + (size_t)i * n_seg * 12);
}

void orc_dt_initial_batch(const fh_problem* pr, int n, double* dt) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; i++) dt[i] = orc_dt_initial(pr + i);
}

void orc_default_params(fh_params* p) {
  p->feas_tol = 1e-9;
  p->dep_tol = 1e-10;
  p->max_nodes = 100000;
  p->max_iters = 2000;
  p->max_work = 0;
  p->share = 1;
  p->mip_gap = 0.0;
  p->deadline_ms = 0.0;
}

size_t orc_sizeof_problem(void) { return sizeof(fh_problem); }
size_t orc_sizeof_result(void) { return sizeof(fh_result); }
