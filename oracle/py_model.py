"""Independent second opinion for the oracle: the UNREDUCED model of faster/src/solverGurobi.cpp.

TEST INFRASTRUCTURE ONLY (same rules as oracle/faster_oracle.c; PARITY UNPINNED — not Gurobi output).

Unlike the C oracle (jerk-space, equalities eliminated, dual active set) this module keeps the
reference's own variables — 12 polynomial coefficients per segment, createVars
(solverGurobi.cpp:70-84) — and writes every constraint row exactly as the reference adds it:
  initial state  setConstraintsX0      :359-380     final state  setConstraintsXf :332-357
  continuity     setDynamicConstraints :499-524     box          setMaxConstraints :390-407
  corridor       setPolytopesConstraints :237-289 with control points getCP0..3 :833-862
  objective      setObjective :113-119  (sum over segments/axes of (6 a)^2)
and solves the fixed-assignment QP with SciPy (SLSQP, then trust-constr as a tie breaker).
Slow; used on small samples in tests/.
"""
import itertools

import numpy as np
from scipy import optimize


def _pos(t, tau, i, N):
    r = np.zeros(12 * N)
    r[12 * t + 0 + i] = tau**3
    r[12 * t + 3 + i] = tau**2
    r[12 * t + 6 + i] = tau
    r[12 * t + 9 + i] = 1
    return r


def _vel(t, tau, i, N):
    r = np.zeros(12 * N)
    r[12 * t + 0 + i] = 3 * tau**2
    r[12 * t + 3 + i] = 2 * tau
    r[12 * t + 6 + i] = 1
    return r


def _acc(t, tau, i, N):
    r = np.zeros(12 * N)
    r[12 * t + 0 + i] = 6 * tau
    r[12 * t + 3 + i] = 2
    return r


def _jerk(t, i, N):
    r = np.zeros(12 * N)
    r[12 * t + 0 + i] = 6
    return r


def _cp(t, k, i, N, dt):
    """Bezier control point k of segment t, axis i (getCP0..3 with the normalised coefficients
    An=a dt^3, Bn=b dt^2, Cn=c dt, Dn=d; :811-862)."""
    r = np.zeros(12 * N)
    if k == 0:
        return _pos(t, 0.0, i, N)
    if k == 3:
        return _pos(t, dt, i, N)
    if k == 1:  # (Cn + 3 Dn)/3
        r[12 * t + 6 + i] = dt / 3.0
        r[12 * t + 9 + i] = 1.0
    if k == 2:  # (Bn + 2 Cn + 3 Dn)/3
        r[12 * t + 3 + i] = dt * dt / 3.0
        r[12 * t + 6 + i] = 2.0 * dt / 3.0
        r[12 * t + 9 + i] = 1.0
    return r


def build(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, assign):
    """Returns (Aeq, beq, Ain, bin) with Aeq c = beq, Ain c <= bin, c in R^{12N}."""
    Aeq, beq, Ain, bin_ = [], [], [], []
    for i in range(3):
        Aeq += [_pos(0, 0, i, N), _vel(0, 0, i, N), _acc(0, 0, i, N)]
        beq += [x0[i], x0[3 + i], x0[6 + i]]
    for i in range(3):
        if force_final:
            Aeq.append(_pos(N - 1, dt, i, N))
            beq.append(xf[i])
        Aeq += [_vel(N - 1, dt, i, N), _acc(N - 1, dt, i, N)]
        beq += [xf[3 + i], xf[6 + i]]
    for t in range(N - 1):
        for i in range(3):
            Aeq += [_pos(t, dt, i, N) - _pos(t + 1, 0, i, N), _vel(t, dt, i, N) - _vel(t + 1, 0, i, N),
                    _acc(t, dt, i, N) - _acc(t + 1, 0, i, N)]
            beq += [0, 0, 0]
    for t in range(N):
        for i in range(3):
            for row, mx in ((_vel(t, 0, i, N), vmax), (_acc(t, 0, i, N), amax), (_jerk(t, i, N), jmax)):
                Ain += [row, -row]
                bin_ += [mx, mx]
    if polys:
        for t in range(N):
            A, b = polys[assign[t]]
            for f in range(len(b)):
                for k in range(4):
                    Ain.append(sum(A[f][i] * _cp(t, k, i, N, dt) for i in range(3)))
                    bin_.append(b[f])
    return np.array(Aeq), np.array(beq), np.array(Ain), np.array(bin_)


def solve_fixed(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, assign, method="SLSQP"):
    """Fixed-assignment QP on the 12N coefficients. Returns (cost, coeff[N,12], max_violation) or None."""
    Aeq, beq, Ain, bin_ = build(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, assign)
    J = np.array([_jerk(t, i, N) for t in range(N) for i in range(3)])
    H = J.T @ J

    def f(c):
        return float(c @ H @ c)

    def g(c):
        return 2 * H @ c

    # feasible-ish start: min-norm solution of the equalities
    c0 = np.linalg.lstsq(Aeq, beq, rcond=None)[0]
    if method == "SLSQP":
        cons = [{"type": "eq", "fun": lambda c: Aeq @ c - beq, "jac": lambda c: Aeq},
                {"type": "ineq", "fun": lambda c: bin_ - Ain @ c, "jac": lambda c: -Ain}]
        r = optimize.minimize(f, c0, jac=g, constraints=cons, method="SLSQP",
                              options={"maxiter": 2000, "ftol": 1e-12})
    else:
        cons = [optimize.LinearConstraint(Aeq, beq, beq), optimize.LinearConstraint(Ain, -np.inf, bin_)]
        r = optimize.minimize(f, c0, jac=g, hess=lambda c: 2 * H, constraints=cons, method="trust-constr",
                              options={"maxiter": 5000, "gtol": 1e-12, "xtol": 1e-14})
    c = r.x
    viol = max(np.max(np.abs(Aeq @ c - beq)), np.max(Ain @ c - bin_))
    ok = r.success or "directional derivative" in str(r.message)  # SLSQP line-search stall at the optimum
    if not ok or viol > 1e-7:
        return None
    return f(c), c.reshape(N, 12), viol


def enumerate_miqp(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, candidates=None):
    """Brute force over assignments with SciPy. Returns (best_cost, best_assign, n_feasible)."""
    P = len(polys)
    best, barg, nfeas = np.inf, None, 0
    it = candidates if candidates is not None else itertools.product(range(P), repeat=N)
    for a in it:
        s = solve_fixed(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, a)
        if s is None:
            continue
        nfeas += 1
        if s[0] < best:
            best, barg = s[0], tuple(a)
    return best, barg, nfeas


def milp_feasible(N, dt, x0, xf, vmax, amax, jmax, force_final, polys, big_m=1.0e3, time_limit=60.0, ineq_slack=0.0):
    """Feasibility of the reference's mixed-integer constraint set at one dt, decided by an independent third-party
    branch-and-cut code (HiGHS through scipy.optimize.milp): 12N coefficients + one binary per (segment, polytope),
    sum_p b[t][p] == 1, and the indicator rows of setPolytopesConstraints (:283-286) written with a big-M.
    Returns True / False (None if HiGHS hits its limits).  Objective: none (feasibility only).
    ineq_slack: added to the right-hand side of every inequality row (box and corridor): > 0 relaxes, < 0 tightens — HiGHS decides with its
    own primal feasibility tolerance (1e-7), the kernels with feas_tol = 1e-9; a caller that finds the two disagreeing asks again with
    +-1e-6 to tell a marginal instance (the verdict flips with the slack) from a real disagreement."""
    from scipy.optimize import Bounds, LinearConstraint, milp

    P = len(polys)
    nc = 12 * N
    nv = nc + N * P
    Aeq, beq, Ain, bin_ = build(N, dt, x0, xf, vmax, amax, jmax, force_final, [], None)
    rows, lo, hi = [], [], []
    for a, b in zip(Aeq, beq):
        rows.append(np.concatenate([a, np.zeros(N * P)])); lo.append(b); hi.append(b)
    for a, b in zip(Ain, bin_):
        rows.append(np.concatenate([a, np.zeros(N * P)])); lo.append(-np.inf); hi.append(b + ineq_slack)
    for t in range(N):
        r = np.zeros(nv)
        r[nc + t * P: nc + (t + 1) * P] = 1.0
        rows.append(r); lo.append(1.0); hi.append(1.0)
        for p, (A, b) in enumerate(polys):
            for f in range(len(b)):
                for k in range(4):
                    r = np.zeros(nv)
                    r[:nc] = sum(A[f][i] * _cp(t, k, i, N, dt) for i in range(3))
                    r[nc + t * P + p] = big_m          # a.cp <= b + M (1 - b_tp)
                    rows.append(r); lo.append(-np.inf); hi.append(b[f] + big_m + ineq_slack)
    cons = LinearConstraint(np.array(rows), np.array(lo), np.array(hi))
    integrality = np.concatenate([np.zeros(nc), np.ones(N * P)])
    bounds = Bounds(np.concatenate([np.full(nc, -1e4), np.zeros(N * P)]), np.concatenate([np.full(nc, 1e4), np.ones(N * P)]))
    res = milp(c=np.zeros(nv), constraints=cons, integrality=integrality, bounds=bounds,
               options={"time_limit": time_limit, "presolve": True})
    if res.status == 0:
        return True
    if res.status == 2:
        return False
    return None


def milp_job(job):
    """(tag, expect_feasible, N, dt, x0, xf, v, a, j, force, polys) -> (tag, expect, verdict, marginal): milp_feasible for one question of a
    test (tests/test_gpu_round6.py), in a worker process of its own.  HiGHS decides with a primal tolerance of 1e-7, the kernels with
    1e-9: on a disagreement it is asked again with every inequality moved by 1e-6 TOWARDS the expected answer — if it then agrees the
    instance is marginal (its verdict hangs on 1e-6 of slack), if not it is a real disagreement."""
    tag, expect, N, dt, x0, xf, v, a, j, force, polys = job
    got = milp_feasible(N, dt, x0, xf, v, a, j, force, polys)
    marginal = False
    if got is not None and got != expect:
        again = milp_feasible(N, dt, x0, xf, v, a, j, force, polys, ineq_slack=1e-6 if expect else -1e-6)
        marginal = again == expect
    return tag, expect, got, marginal
