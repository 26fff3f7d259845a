"""Self-certification of the CPU oracle on synthetic corridors (the reference gives no expected outputs):
exhaustive enumeration of all P^N assignments, an independent SciPy solve of the unreduced model, and invariances."""
import numpy as np
import pytest

from faster_amd import abi, corridor


def polys_of(pr, faces):
    out = []
    fb = int(pr["face_begin"])
    for p in range(int(pr["n_poly"])):
        f0, f1 = fb + pr["face_off"][p], fb + pr["face_off"][p + 1]
        out.append((faces["a"][f0:f1].copy(), faces["b"][f0:f1].copy()))
    return out


def test_branch_and_bound_equals_exhaustive_enumeration(oracle):
    """All P^N assignments (N=5, P<=3: up to 243 QPs per problem) vs. branch and bound, at the accepted dt and at the
    last rejected dt (where every assignment must be infeasible)."""
    pr, faces, _ = corridor.whole_batch(40, seed=101, n_seg=5, p_choices=(2, 3), speed=3.5, lateral=1.0)
    res = oracle.solve_batch(pr, faces)
    checked = 0
    for i in range(len(pr)):
        r = res[i]
        if not r["solved"]:
            continue
        nfeas, bf = oracle.bruteforce_dt(pr[i], faces, r["dt"])
        assert nfeas >= 1
        assert bf["cost"] == pytest.approx(r["cost"], rel=1e-9, abs=1e-9)
        np.testing.assert_allclose(bf["coeff"], r["coeff"], atol=1e-7)
        if r["trials"] > 1:
            dt_prev = (r["factor"] - pr[i]["f_inc"]) * (r["dt"] / r["factor"])
            nprev, _ = oracle.bruteforce_dt(pr[i], faces, dt_prev)
            assert nprev == 0
        checked += 1
    assert checked >= 20


def test_safe_problems_equal_exhaustive_enumeration(oracle):
    pr, faces, _ = corridor.safe_batch(30, seed=102, n_seg=5, p_choices=(2, 3), speed=3.0)
    res = oracle.solve_batch(pr, faces)
    assert res["solved"].mean() > 0.5
    for i in np.nonzero(res["solved"])[0]:
        nfeas, bf = oracle.bruteforce_dt(pr[i], faces, res[i]["dt"])
        assert bf["cost"] == pytest.approx(res[i]["cost"], rel=1e-9, abs=1e-9)


def test_fast_safe_problems_every_trial_equals_exhaustive_enumeration(oracle):
    """Safe problems that start fast (the root relaxation of a trial ends outside the corridor, so the branch and bound takes the
    earliest violated segment below the root — or at the root, when it overshoots by more than 1.2 braking distances): EVERY trial
    of the factor window, feasible or not, against all P^N assignments (N = 5, P = 3: 243 QPs per trial)."""
    pr, faces, verts = corridor.safe_batch(32, seed=104, n_seg=5, p_choices=(2, 3))
    faces = faces.copy()
    faces["b"] -= 0.4  # (unit normals: every face 0.4 m closer; the start stays inside)
    u = verts[:, 1] - verts[:, 0]
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pr["x0"][:, 3:6], pr["x0"][:, 6:9] = 4.8 * u, 2.0 * u  # fast and still accelerating along the corridor
    res = oracle.solve_batch(pr, faces)
    assert 0.1 < res["solved"].mean() < 1.0   # a mix: some of these cannot stop inside their corridor for any factor
    infeasible_trials = feasible_trials = 0
    for i in range(len(pr)):
        base = oracle.dt_initial(pr[i])
        base = max(base, 2 * float(pr[i]["dc"]))
        f = float(pr[i]["f_init"])
        for _ in range(4):
            dt = f * base
            st, r = oracle.miqp_dt(pr[i], faces, dt)
            nfeas, bf = oracle.bruteforce_dt(pr[i], faces, dt)
            if nfeas == 0:
                assert st == abi.FH_ST_INFEASIBLE, (i, f)
                infeasible_trials += 1
            else:
                assert st == abi.FH_ST_OPTIMAL, (i, f)
                assert bf["cost"] == pytest.approx(r["cost"], rel=1e-9, abs=1e-9)
                feasible_trials += 1
            f += float(pr[i]["f_inc"])
    assert infeasible_trials >= 20 and feasible_trials >= 20


def test_scipy_unreduced_model_agrees_on_random_corridors(oracle):
    """SLSQP on the 12N-coefficient model with the oracle's assignment must not find a better or different optimum."""
    from oracle import py_model

    pr, faces, _ = corridor.whole_batch(12, seed=103, n_seg=6, p_choices=(1, 2, 3))
    res = oracle.solve_batch(pr, faces)
    done = 0
    for i in np.nonzero(res["solved"])[0][:8]:
        p, r = pr[i], res[i]
        N = int(p["n_seg"])
        s = py_model.solve_fixed(N, float(r["dt"]), p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]),
                                 bool(p["force_final_pos"]), polys_of(p, faces), [int(a) for a in r["assign"][:N]])
        assert s is not None
        assert s[0] == pytest.approx(r["cost"], rel=1e-7, abs=1e-8)
        np.testing.assert_allclose(s[1], r["coeff"][:N], atol=5e-6)
        done += 1
    assert done >= 4


def test_solution_satisfies_the_reference_model(oracle):
    """Every reported trajectory satisfies the rows the reference adds (solverGurobi.cpp:332-407, :499-524, :237-289)."""
    pr, faces, _ = corridor.whole_batch(64, seed=104, p_choices=(2, 3, 4, 5))
    res = oracle.solve_batch(pr, faces)
    for i in np.nonzero(res["solved"])[0]:
        p, r = pr[i], res[i]
        N, h = int(p["n_seg"]), float(r["dt"])
        c = r["coeff"][:N]
        a, b, cc, d = c[:, 0:3], c[:, 3:6], c[:, 6:9], c[:, 9:12]
        pos_end = a * h**3 + b * h**2 + cc * h + d
        vel_end = 3 * a * h**2 + 2 * b * h + cc
        acc_end = 6 * a * h + 2 * b
        np.testing.assert_allclose(d[0], p["x0"][0:3], atol=1e-12)
        np.testing.assert_allclose(cc[0], p["x0"][3:6], atol=1e-12)
        np.testing.assert_allclose(2 * b[0], p["x0"][6:9], atol=1e-12)
        np.testing.assert_allclose(pos_end[:-1], d[1:], atol=1e-9)
        np.testing.assert_allclose(vel_end[:-1], cc[1:], atol=1e-9)
        np.testing.assert_allclose(acc_end[:-1], 2 * b[1:], atol=1e-9)
        np.testing.assert_allclose(pos_end[-1], p["xf"][0:3], atol=1e-8)
        np.testing.assert_allclose(vel_end[-1], 0, atol=1e-8)
        np.testing.assert_allclose(acc_end[-1], 0, atol=1e-8)
        assert np.all(np.abs(cc) <= p["v_max"] + 1e-8) and np.all(np.abs(2 * b) <= p["a_max"] + 1e-8)
        assert np.all(np.abs(6 * a) <= p["j_max"] + 1e-8)
        assert r["cost"] == pytest.approx(float(np.sum((6 * a) ** 2)), rel=1e-12)
        fb = int(p["face_begin"])
        for t in range(N):
            q = int(r["assign"][t])
            f0, f1 = fb + p["face_off"][q], fb + p["face_off"][q + 1]
            for cp in (d[t], d[t] + cc[t] * h / 3, d[t] + 2 * cc[t] * h / 3 + b[t] * h * h / 3, pos_end[t]):
                assert np.max(faces["a"][f0:f1] @ cp - faces["b"][f0:f1]) <= 1e-7


def test_invariances(oracle):
    """Translation of the whole scene, permutation and duplication of faces, and relabelling of polytopes leave
    feasibility, factor and cost unchanged."""
    pr, faces, _ = corridor.whole_batch(24, seed=105, p_choices=(2, 3))
    base = oracle.solve_batch(pr, faces)
    shift = np.array([3.25, -7.5, 0.0])
    pr2, faces2 = pr.copy(), faces.copy()
    pr2["x0"][:, 0:3] += shift
    pr2["xf"][:, 0:3] += shift
    faces2["b"] += faces2["a"] @ shift
    r2 = oracle.solve_batch(pr2, faces2)
    assert np.array_equal(r2["solved"], base["solved"]) and np.array_equal(r2["factor"], base["factor"])
    np.testing.assert_allclose(r2["cost"], base["cost"], rtol=1e-7, atol=1e-9)
    # reverse the face order inside every polytope and duplicate the first face
    rng = np.random.default_rng(0)
    batches = []
    for i in range(len(pr)):
        ps = polys_of(pr[i], faces)
        new = []
        for A, b in ps:
            perm = rng.permutation(len(b))
            new.append((np.vstack([A[perm], A[perm][:1]]), np.concatenate([b[perm], b[perm][:1]])))
        new = new[::-1]  # relabel polytopes
        f, off = abi.pack_faces(new)
        q = pr[i: i + 1].copy()
        q["face_begin"] = 0
        q["face_off"][0, : len(off)] = off
        q["face_off"][0, len(off):] = off[-1]
        batches.append((q, f))
    pr3, faces3 = corridor.concat(batches)
    r3 = oracle.solve_batch(pr3, faces3)
    assert np.array_equal(r3["solved"], base["solved"]) and np.array_equal(r3["factor"], base["factor"])
    np.testing.assert_allclose(r3["cost"], base["cost"], rtol=1e-7, atol=1e-9)
    # (the polytope index reported for a segment that fits several overlapping polytopes may legitimately differ)
    np.testing.assert_allclose(r3["coeff"], base["coeff"], atol=1e-6)


def test_cost_does_not_increase_with_more_time_in_free_space(oracle):
    """Without a corridor the optimal jerk cost is non-increasing in dt (more time is never worse)."""
    pr, faces, _ = corridor.whole_batch(8, seed=106)
    pr = pr.copy()
    pr["n_poly"] = 0
    res = oracle.solve_batch(pr, faces)
    for i in np.nonzero(res["solved"])[0]:
        costs = []
        for f in (1.0, 1.25, 1.5, 2.0):
            st, r = oracle.miqp_dt(pr[i], faces, f * res[i]["dt"])
            assert st == abi.FH_ST_OPTIMAL
            costs.append(r["cost"])
        assert all(a >= b - 1e-9 for a, b in zip(costs, costs[1:]))


def test_first_feasible_factor_against_highs(oracle):
    """Feasibility flag / factor_that_worked_ pinned by a third-party mixed-integer code (HiGHS via scipy.optimize.milp) on
    the reference's own variables with big-M indicator rows: feasible at the accepted factor, infeasible one step earlier."""
    from oracle import py_model

    pr, faces, _ = corridor.whole_batch(10, seed=107, n_seg=5, p_choices=(2, 3), speed=3.0, lateral=0.8)
    res = oracle.solve_batch(pr, faces)
    checked = 0
    for i in np.nonzero(res["solved"])[0][:6]:
        p, r = pr[i], res[i]
        args = (int(p["n_seg"]),)
        common = (p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), bool(p["force_final_pos"]), polys_of(p, faces))
        assert py_model.milp_feasible(args[0], float(r["dt"]), *common) is True
        if r["trials"] > 1:
            dt_prev = (r["factor"] - p["f_inc"]) * (r["dt"] / r["factor"])
            assert py_model.milp_feasible(args[0], float(dt_prev), *common) is False
        checked += 1
    assert checked >= 3


def test_child_bound_variant_gives_the_same_results(oracle, tmp_path):
    """The experimental variant (-DORC_PARENT_BOUND: children whose one-row dual bound at the parent already loses against the
    incumbent are not visited; the kernels' -DFH_PARENT_BOUND) only skips nodes that hold no better leaf: same solved / trials /
    factor / cost / trajectory, never more nodes, on whole and on fast safe problems."""
    import ctypes
    import os
    import subprocess

    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    so = str(tmp_path / "liboracle_pb.so")
    subprocess.check_call(["gcc", "-O2", "-march=native", "-fPIC", "-fopenmp", "-ffp-contract=off", "-DORC_PARENT_BOUND", "-shared", "-o", so,
                           os.path.join(here, "faster_oracle.c"), "-lm"])
    L = ctypes.CDLL(so)
    L.orc_solve_batch_mt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]

    def variant(pr, faces):
        pr, faces = np.ascontiguousarray(pr), np.ascontiguousarray(faces)
        res = np.zeros(len(pr), dtype=abi.result_dtype)
        par = abi.default_params()
        L.orc_solve_batch_mt(abi.ptr(pr), abi.ptr(faces), abi.ptr(par.reshape(1)), len(pr), abi.ptr(res), 0)
        return res

    whole, wf, _ = corridor.whole_batch(256, seed=105, n_seg=10, p_choices=(3, 4, 5, 6))
    safe, sf, verts = corridor.safe_batch(256, seed=106, n_seg=8, p_choices=(3, 4))
    sf = sf.copy()
    sf["b"] -= 0.4
    u = verts[:, 1] - verts[:, 0]
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    safe["x0"][:, 3:6], safe["x0"][:, 6:9] = 4.8 * u, 2.0 * u
    fewer = 0
    for pr, faces in ((whole, wf), (safe, sf)):
        ref, got = oracle.solve_batch(pr, faces), variant(pr, faces)
        for f in ("solved", "trials", "factor", "dt", "status"):
            assert np.array_equal(ref[f], got[f]), f
        ok = ref["solved"] == 1
        np.testing.assert_allclose(got["cost"][ok], ref["cost"][ok], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(got["coeff"][ok], ref["coeff"][ok], rtol=0, atol=1e-7)
        assert np.all(got["nodes"] <= ref["nodes"])
        fewer += int((got["nodes"] < ref["nodes"]).sum())
    assert fewer >= 64
