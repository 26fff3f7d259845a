"""Pins the CPU oracle (oracle/faster_oracle.c) to every fixture the reference holds for the solver path.

The reference has NO expected outputs for SolverGurobi: the only fixture is the hard-coded
corridor of faster/other/gurobi_continuous.cpp (inputs).  The expected values below come from the survey's
independent SciPy computation and from oracle/py_model.py run here — parity unpinned.
"""
import numpy as np
import pytest

from faster_amd import abi, corridor


def _case(fx, c, **kw):
    return corridor.fixture_problem(fx, c["N"], c["vaj"], c["force_final"], c["polys"], c["x0"], c["xf"], **kw)


@pytest.mark.parametrize("name", ["KA-1", "KA-2", "KA-3", "KA-4"])
def test_known_answer(oracle, fixture_corridor, known_answers, name):
    c = known_answers["cases"][name]
    pr, faces = _case(fixture_corridor, c)
    r = oracle.solve_batch(pr, faces)[0]
    assert r["solved"] == 1 and r["status"] == abi.FH_ST_OPTIMAL
    assert r["trials"] == c["trials"] and r["factor"] == c["factor"]
    assert list(r["assign"][: c["N"]]) == c["assign"]
    if "dt_init" in c:  # float-cast semantics of getDTInitial (solverGurobi.cpp:662-670,751)
        assert oracle.dt_initial(pr) == c["dt_init"]
        assert r["dt"] == c["dt"]
        assert r["cost"] == pytest.approx(c["cost"], rel=1e-10)
    else:
        # the survey's dt_init is one float ulp off the reference's float/int division; compare at its dt
        assert r["dt"] == pytest.approx(c["dt"], rel=1e-7)
        st, r2 = oracle.miqp_dt(pr, faces, c["dt"])
        assert st == abi.FH_ST_OPTIMAL
        assert r2["cost"] == pytest.approx(c["cost"], rel=c.get("cost_tol", 1e-10))
        assert list(r2["assign"][: c["N"]]) == c["assign"]


def test_dt_initial_is_float_division(oracle, fixture_corridor, known_answers):
    """dt_initial = max(float...) / N_ is a float/int division (solverGurobi.cpp:751)."""
    c = known_answers["cases"]["KA-2"]
    pr, _ = _case(fixture_corridor, c)
    expect = float(np.float32(np.sqrt(2 * 9.0 / 5.0)) / np.float32(6))  # t_a on x dominates
    assert oracle.dt_initial(pr) == expect


def test_runners_up_and_first_feasible_factor(oracle, fixture_corridor, known_answers):
    c = known_answers["cases"]["KA-1"]
    pr, faces = _case(fixture_corridor, c)
    for cost, assign in c["runners_up"]:
        st, r = oracle.miqp_dt(pr, faces, c["dt"], assign=assign)
        assert st == abi.FH_ST_OPTIMAL
        assert r["cost"] == pytest.approx(cost, rel=2e-8)
    # factors 1 and 2 are infeasible for every assignment (App. B)
    dti = oracle.dt_initial(pr)
    for f in (1.0, 2.0):
        st, _ = oracle.miqp_dt(pr, faces, f * dti)
        assert st == abi.FH_ST_INFEASIBLE


def test_feasible_assignment_counts(oracle, fixture_corridor, known_answers):
    """Brute force over ALL P^N assignments agrees with branch and bound and with the survey's counts."""
    c = known_answers["cases"]["KA-2"]
    pr, faces = _case(fixture_corridor, c)
    nfeas, r = oracle.bruteforce_dt(pr, faces, c["dt"])
    assert nfeas == c["n_feasible"]
    assert r["cost"] == pytest.approx(c["cost"], rel=1e-7)
    assert list(r["assign"][:6]) == c["assign"]


@pytest.mark.parametrize("name", ["KA-1", "KA-3"])
def test_unreduced_scipy_model_agrees(oracle, fixture_corridor, known_answers, name):
    """Second opinion: SLSQP on the reference's own 12N-coefficient formulation (oracle/py_model.py)."""
    from oracle import py_model

    c = known_answers["cases"][name]
    polys = [(np.array(fixture_corridor["polytopes"][p]["A"]), np.array(fixture_corridor["polytopes"][p]["b"])) for p in c["polys"]]
    remap = {p: i for i, p in enumerate(c["polys"])}
    s = py_model.solve_fixed(c["N"], c["dt"], c["x0"], c["xf"], *c["vaj"], bool(c["force_final"]), polys,
                             [remap.get(a, a) for a in c["assign"]])
    assert s is not None
    pr, faces = _case(fixture_corridor, c)
    r = oracle.solve_batch(pr, faces)[0]
    assert s[0] == pytest.approx(r["cost"], rel=1e-9)
    np.testing.assert_allclose(s[1], r["coeff"][: c["N"]], atol=2e-6)


def test_sampling_matches_fillx_semantics(oracle, fixture_corridor, known_answers):
    """resetX/fillX (solverGurobi.cpp:382-388,122-168): first sample at t=DC, last sample has zero vel/acc/jerk."""
    c = known_answers["cases"]["KA-3"]
    pr, faces = _case(fixture_corridor, c)
    r = oracle.solve_batch(pr, faces)[0]
    X = oracle.sample(pr[0], r)
    n = max(2, int(c["N"] * r["dt"] / 0.01))
    assert X.shape[0] == n
    co = r["coeff"][0]
    tau = 0.01
    np.testing.assert_allclose(X[0]["pos"], co[0:3] * tau**3 + co[3:6] * tau**2 + co[6:9] * tau + co[9:12], rtol=0, atol=1e-14)
    assert np.all(X[-1]["vel"] == 0) and np.all(X[-1]["accel"] == 0) and np.all(X[-1]["jerk"] == 0)
    assert np.all(X[-2]["jerk"] == 6 * r["coeff"][c["N"] - 1][0:3])


def test_fixture_first_feasible_factor_against_highs(oracle, fixture_corridor, known_answers):
    """KA-1 on the reference's corridor: HiGHS (scipy.optimize.milp, big-M indicators on the unreduced variables) finds the
    constraint set infeasible at factors 1 and 2 and feasible at factor 3 — the `trials_`/`factor_that_worked_` the oracle reports."""
    from oracle import py_model

    c = known_answers["cases"]["KA-1"]
    polys = [(np.array(fixture_corridor["polytopes"][p]["A"]), np.array(fixture_corridor["polytopes"][p]["b"])) for p in c["polys"]]
    dti = c["dt_init"]
    args = (c["x0"], c["xf"], *c["vaj"], True, polys)
    assert py_model.milp_feasible(c["N"], 2 * dti, *args) is False
    assert py_model.milp_feasible(c["N"], 3 * dti, *args) is True


def test_control_points_two_routes_agree(oracle, fixture_corridor, known_answers):
    """getCP0..3 (solverGurobi.cpp:833-862): the product's host conversion fh_control_points (the reference's literal expressions) and
    the oracle's jerk-space route give the same points on the known answers, and every point lies in its segment's polytope."""
    from faster_amd import capi

    batches = []
    for name in ("KA-1", "KA-2", "KA-3", "KA-4"):
        c = known_answers["cases"][name]
        batches.append(corridor.fixture_problem(fixture_corridor, c["N"], c["vaj"], c["force_final"], c["polys"], c["x0"], c["xf"]))
    for (pr, faces) in batches:
        res = oracle.solve_batch(pr, faces)
        assert res["solved"][0] == 1
        N = int(pr[0]["n_seg"])
        a, b = capi.control_points(res, N), oracle.control_points(res, N)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(a[0, 0, 0], pr[0]["x0"][:3])          # cp0 of the first segment is the start
        for t in range(N):
            q = int(res["assign"][0][t])
            f0, f1 = pr[0]["face_begin"] + pr[0]["face_off"][q], pr[0]["face_begin"] + pr[0]["face_off"][q + 1]
            assert np.max(faces["a"][f0:f1] @ a[0, t].T - faces["b"][f0:f1, None]) <= 1e-6
        if N > 1:
            np.testing.assert_allclose(a[0, :-1, 3], a[0, 1:, 0], rtol=0, atol=1e-9)   # C0: cp3 of a segment is cp0 of the next


def test_time_allocation_when_start_and_goal_coincide_on_an_axis(oracle):
    """[r6] getDTInitial (solverGurobi.cpp:659-759) with xf == x0 exactly on an axis: the constant term of that axis' cubic (:691-693)
    is zero, one root is exactly 0 and is not positive (MinPositiveElement, solverGurobi_utils.hpp:19-32); what is left are the roots of
    (j/6) t^2 + (a0/2) t + v0.  The oracle's convention (real_roots_cubic) against an independent evaluation: numpy's companion-matrix
    roots of the QUADRATIC (the zero root divided out by hand), the reference's float casts, the maximum over the nine values."""
    rng = np.random.default_rng(66)
    n = 4096
    pr = np.zeros(n, dtype=abi.problem_dtype)
    pr["n_seg"] = rng.choice([6, 10, 15], n)
    pr["dc"], pr["v_max"], pr["a_max"], pr["j_max"] = 0.01, 5.0, 5.0, 8.0
    x0, xf = np.zeros((n, 9)), np.zeros((n, 9))
    x0[:, :3] = rng.uniform(-10, 10, (n, 3))
    xf[:, :3] = x0[:, :3] + rng.uniform(-6, 6, (n, 3))
    same = rng.random((n, 3)) < 0.6
    same[:, 0] |= ~same.any(axis=1)
    xf[:, :3] = np.where(same, x0[:, :3], xf[:, :3])
    x0[:, 3:6] = rng.uniform(-4, 4, (n, 3)) * (rng.random((n, 3)) < 0.8)
    x0[:, 6:9] = rng.uniform(-4, 4, (n, 3)) * (rng.random((n, 3)) < 0.8)
    pr["x0"], pr["xf"] = x0, xf
    got = oracle.dt_initial_batch(pr)

    def min_pos(roots):
        r = [float(z.real) for z in roots if abs(z.imag) < 1e-12 and z.real > 0]
        return min(r) if r else 0.0

    worst = 0.0
    for i in range(n):
        vals = []
        for a in range(3):
            dx = xf[i, a] - x0[i, a]
            sg = np.copysign(1.0, dx)
            j, acc = float(np.float32(sg * 8.0)), float(np.float32(sg * 5.0))
            v0, a0 = float(np.float32(x0[i, 3 + a])), float(np.float32(x0[i, 6 + a]))
            vals.append(np.float32(abs(dx) / 5.0))
            if dx == 0.0:
                cub = min_pos(np.roots([j / 6.0, a0 / 2.0, v0])) if (a0 != 0 or v0 != 0) else 0.0
            else:
                cub = min_pos(np.roots([j / 6.0, a0 / 2.0, v0, -dx]))
            vals.append(np.float32(cub))
            vals.append(np.float32(min_pos(np.roots([0.5 * acc, v0, -dx]))))
        want = float(np.float32(max(vals)) / np.float32(pr["n_seg"][i]))
        worst = max(worst, abs(got[i] - want) / max(want, 1e-12))
        assert got[i] == pytest.approx(want, rel=3e-7, abs=1e-12), (i, got[i], want, x0[i], xf[i])
    # hand cases: at rest with nowhere to go -> 0; moving away on an idle axis: the positive root of the quadratic
    p = pr[:2].copy()
    p["x0"][:], p["xf"][:] = 0.0, 0.0
    p["x0"][1, 3], p["x0"][1, 6] = 2.0, -3.0     # v0 = 2, a0 = -3, dx = 0 -> copysign(1, 0) = +1: (8/6) t^2 - 1.5 t + 2 has no real root
    p["n_seg"] = 10
    d = oracle.dt_initial_batch(p)
    assert d[0] == 0.0 and d[1] == 0.0
    p["x0"][1, 3] = -2.0                          # v0 = -2: (8/6) t^2 - 1.5 t - 2 = 0 -> t = (1.5 + sqrt(2.25 + 32/3)) / (8/3) = 1.9106...
    d = oracle.dt_initial_batch(p)
    assert d[1] == pytest.approx((1.5 + np.sqrt(2.25 + 32.0 / 3.0)) / (8.0 / 3.0) / 10.0, rel=2e-7)
