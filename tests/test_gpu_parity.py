"""GPU parity: the HIP path (through the C ABI, faster_amd/libfasterhip.so) against the CPU oracle.

Bars: feasibility flag, trials_ and factor_that_worked_ exact; cost 1e-7 relative
(north-star bar: 1e-4); polynomial coefficients 1e-6 absolute.  Parity is UNPINNED with respect to Gurobi
(absent); the oracle itself is pinned in tests/test_oracle_*.py.
"""
import numpy as np
import pytest

from faster_amd import abi, capi, corridor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    # PyTorch-ROCm bundles its own HIP runtime: when a process uses both, torch must be imported BEFORE
    # libfasterhip.so is loaded so that both share one runtime.
    import torch  # noqa: F401

    from faster_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def compare(got, ref, n_seg=None, cost_rtol=1e-7, coeff_atol=1e-6):
    assert np.array_equal(got["status"] == abi.FH_ST_BAD_INPUT, ref["status"] == abi.FH_ST_BAD_INPUT)
    assert np.array_equal(got["solved"], ref["solved"]), np.nonzero(got["solved"] != ref["solved"])
    assert np.array_equal(got["trials"], ref["trials"])
    assert np.array_equal(got["factor"], ref["factor"])
    assert np.array_equal(got["dt"], ref["dt"])  # same float-cast time allocation, bit for bit
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["solved"] == 1
    np.testing.assert_allclose(got["cost"][ok], ref["cost"][ok], rtol=cost_rtol, atol=1e-9)
    np.testing.assert_allclose(got["coeff"][ok], ref["coeff"][ok], rtol=0, atol=coeff_atol)
    return ok


def check_assignment_valid(pr, faces, res, tol=1e-6):
    """Every segment's 4 Bezier control points lie in the polytope the result assigns it to."""
    for i in np.nonzero(res["solved"] == 1)[0]:
        p, r = pr[i], res[i]
        h, N = r["dt"], int(p["n_seg"])
        for t in range(N):
            if p["n_poly"] == 0:
                assert r["assign"][t] == -1
                continue
            a, b, c, d = (r["coeff"][t][3 * k: 3 * k + 3] for k in range(4))
            cps = [d, d + c * h / 3, d + 2 * c * h / 3 + b * h * h / 3, a * h**3 + b * h**2 + c * h + d]
            q = int(r["assign"][t])
            assert 0 <= q < p["n_poly"]
            f0, f1 = p["face_begin"] + p["face_off"][q], p["face_begin"] + p["face_off"][q + 1]
            for cp in cps:
                assert np.max(faces["a"][f0:f1] @ cp - faces["b"][f0:f1]) <= tol


def test_known_answers_on_gpu(ctx, oracle, fixture_corridor, known_answers):
    batches = []
    for name in ("KA-1", "KA-2", "KA-3", "KA-4"):
        c = known_answers["cases"][name]
        batches.append(corridor.fixture_problem(fixture_corridor, c["N"], c["vaj"], c["force_final"], c["polys"], c["x0"], c["xf"]))
    pr, faces = corridor.concat(batches)
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)
    assert got["cost"][0] == pytest.approx(23.869158339522752, rel=1e-9)
    assert list(got["assign"][0][:10]) == [0, 0, 0, 0, 0, 0, 2, 2, 2, 2]
    assert got["cost"][2] == pytest.approx(22.019951865206, rel=1e-9)


@pytest.mark.parametrize("name,make", [
    ("C2 safe pure-QP N=6 P=1", lambda: corridor.safe_batch(1024, seed=1)),
    ("C3 whole N=10 P<=4", lambda: corridor.whole_batch(1024, seed=2)),
    ("C4 whole N=10 P<=6", lambda: corridor.whole_batch(512, seed=3, p_choices=(2, 3, 4, 5, 6))),
    ("C5 whole N=15 P<=8", lambda: corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))),
    ("safe N=10 P<=3 multi-polytope", lambda: corridor.safe_batch(256, seed=7, n_seg=10, p_choices=(1, 2, 3))),
])
def test_parity_synthetic(ctx, oracle, name, make):
    pr, faces, _ = make()
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    ok = compare(got, ref)
    assert ok.mean() > 0.5, "generator should give mostly feasible problems"
    check_assignment_valid(pr, faces, got)


def test_edge_cases(ctx, oracle):
    pr, faces, _ = corridor.whole_batch(16, seed=21)
    pr = pr.copy()
    pr["n_poly"][0] = 0                      # no corridor at all
    pr["x0"][1, 3] = 7.0                     # initial speed above v_max: infeasible for every factor
    pr["n_seg"][2] = 0                       # bad input
    pr["n_seg"][3] = abi.FH_MAX_SEG + 1      # bad input
    pr["f_final"][4] = 0.5                   # empty factor window: zero trials
    pr["xf"][5, 0:3] = pr["x0"][5, 0:3]      # goal == start
    pr["f_inc"][6] = 0.0                     # bad input
    pr["x0"][7, 0] = np.nan                  # bad input
    pr["n_seg"][8] = 1
    pr["n_seg"][9] = 3
    pr["f_init"][10], pr["f_final"][10], pr["f_inc"][10] = 1.5, 4.0, 0.25   # non-integer window, accumulated in double
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)
    assert got["status"][2] == abi.FH_ST_BAD_INPUT and got["status"][6] == abi.FH_ST_BAD_INPUT
    assert got["trials"][4] == 0 and got["solved"][4] == 0
    assert got["solved"][1] == 0


def test_empty_batch(ctx):
    pr = abi.make_problems(0)
    faces = np.zeros(0, dtype=abi.face_dtype)
    assert ctx.solve_batch(pr, faces).shape == (0,)


def test_sampling_parity(ctx, oracle):
    pr, faces, _ = corridor.whole_batch(64, seed=31)
    res = ctx.solve_batch(pr, faces)
    cap = 4096
    states, counts = ctx.sample_batch(pr, res, cap)
    for i in range(len(pr)):
        ref = oracle.sample(pr[i], res[i])
        assert counts[i] == ref.shape[0]
        m = min(cap, ref.shape[0])
        for fld in ("pos", "vel", "accel", "jerk"):
            np.testing.assert_allclose(states[i, :m][fld], ref[:m][fld], rtol=0, atol=1e-11)
        if m:
            assert np.all(states[i, m - 1]["vel"] == 0) and np.all(states[i, m - 1]["jerk"] == 0)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def test_pair_pipeline_device_pointers(ctx, oracle):
    """C4 pipeline with device-resident buffers: whole solve -> fh_pair_glue_device -> safe solve, against the
    oracle driven through the host restatement of the hand-off (oracle/pair_glue.py)."""
    import torch

    from oracle import pair_glue

    n, N = 384, 10
    whole, faces, _ = corridor.whole_batch(n, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    tmpl = corridor.safe_templates(whole)
    max_faces = int(whole["face_off"][np.arange(n), whole["n_poly"]].max())
    d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(tmpl)
    d_sfaces = torch.zeros_like(d_faces)
    d_wres = torch.zeros(n * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_sres = torch.zeros_like(d_wres)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.timing_reset()
    ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), n, N, max_faces, d_wres.data_ptr())
    ctx.pair_glue_device(d_whole.data_ptr(), d_wres.data_ptr(), d_faces.data_ptr(), n, 0.5, 0.2, 3, d_safe.data_ptr(), d_sfaces.data_ptr())
    ctx.solve_batch_device(d_safe.data_ptr(), d_sfaces.data_ptr(), n, N, max_faces, d_sres.data_ptr())
    ctx.sync()
    assert len(ctx.timing_read()) == 2 and ctx.last_kernel_ms() > 0
    ctx.set_stream(0)
    wres = d_wres.cpu().numpy().view(abi.result_dtype)
    sres = d_sres.cpu().numpy().view(abi.result_dtype)
    safe = d_safe.cpu().numpy().view(abi.problem_dtype)
    sfaces = d_sfaces.cpu().numpy().view(abi.face_dtype)

    wref = oracle.solve_batch(whole, faces)
    compare(wres, wref)
    # hand-off: same R, same corridor selection (inputs: the GPU's whole results, so that rounding cannot fork it)
    safe_ref, sfaces_ref = pair_glue.glue(whole, wres, faces, tmpl, 0.5, 0.2, 3)
    assert np.array_equal(safe["n_seg"], safe_ref["n_seg"])
    assert np.array_equal(safe["n_poly"], safe_ref["n_poly"])
    assert np.array_equal(safe["face_off"], safe_ref["face_off"])
    np.testing.assert_allclose(safe["x0"], safe_ref["x0"], rtol=0, atol=1e-11)
    live = safe["n_seg"] > 0
    for i in np.nonzero(live)[0]:
        f0 = safe["face_begin"][i]
        f1 = f0 + safe["face_off"][i][safe["n_poly"][i]]
        np.testing.assert_allclose(sfaces["a"][f0:f1], sfaces_ref["a"][f0:f1], rtol=0, atol=0)
        np.testing.assert_allclose(sfaces["b"][f0:f1], sfaces_ref["b"][f0:f1], rtol=0, atol=1e-14)
    sref = oracle.solve_batch(safe, sfaces)
    compare(sres, sref)
    check_assignment_valid(safe, sfaces, sres)
    assert 0.3 < sres["solved"].mean() <= 1.0


def test_maximum_sizes(ctx, oracle):
    """N = FH_MAX_SEG = 16 segments, P = FH_MAX_POLY = 8 polytopes (the NSEG=16 kernel instantiation)."""
    pr, faces, _ = corridor.whole_batch(48, seed=41, n_seg=abi.FH_MAX_SEG, p_choices=(abi.FH_MAX_POLY,))
    got = ctx.solve_batch(pr, faces)  # (with work sharing: helper workgroups take over subtrees of the 48 problems)
    ref = oracle.solve_batch(pr, faces)
    ok = compare(got, ref)
    assert ok.sum() >= 24
    check_assignment_valid(pr, faces, got)
    # same exact method and branching rule => the same branch-and-bound tree on these whole problems when one wavefront explores
    # a tree alone, except that the kernel skips siblings it can prove infeasible from a child's Farkas certificate
    # (conflict-directed backjumping) while the oracle enumerates them: never more nodes, and most trees node for node (a
    # regression check on the node-state snapshots of the NVP = 48 instantiation; with sharing, pruning depends on when another
    # wavefront's incumbent arrives, so only the results are compared there)
    # The child bound (fh_sched.child_bound, on by default) skips children that cannot hold a better leaf: switched off, the trees are
    # the oracle's; switched on, never larger — and the results are the same bit for bit in all three runs.
    solo = capi.Context(0)
    par = abi.default_params()
    par["share"] = 0
    solo.set_params(par)
    bounded = solo.solve_batch(pr, faces)
    solo.set_sched(child_bound=0)
    alone = solo.solve_batch(pr, faces)
    solo.close()
    assert alone["nodes"].max() > 50 and np.all(alone["nodes"] <= ref["nodes"]) and (alone["nodes"] == ref["nodes"]).mean() > 0.5
    assert np.all(bounded["nodes"] <= alone["nodes"]) and bounded["nodes"].sum() < 0.8 * alone["nodes"].sum()
    for f in ("solved", "trials", "status", "factor", "dt", "cost", "coeff", "assign"):
        assert np.array_equal(alone[f], got[f]), f
        assert np.array_equal(bounded[f], got[f]), f


def test_mixed_sizes_in_one_batch(ctx, oracle):
    """Ragged batch: different N, P, face counts and whole/safe problems in one launch."""
    parts = [corridor.whole_batch(40, seed=51, n_seg=4, p_choices=(1, 2))[:2],
             corridor.safe_batch(40, seed=52, n_seg=7, p_choices=(1, 2, 3))[:2],
             corridor.whole_batch(40, seed=53, n_seg=10, p_choices=(5, 6))[:2],
             corridor.whole_batch(20, seed=54, n_seg=13, p_choices=(3, 7))[:2]]
    pr, faces = corridor.concat(parts)
    perm = np.random.default_rng(5).permutation(len(pr))
    pr = pr[perm]
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)


def test_many_faces_and_degenerate_rows(ctx, oracle):
    """Polytopes padded with redundant faces up to FH_MAX_FACES_POLY, duplicated faces, and zero-normal rows
    (0 <= b: harmless when b >= 0, excludes the polytope when b < 0)."""
    pr, faces, _ = corridor.whole_batch(24, seed=61, n_seg=8, p_choices=(2, 3))
    rng = np.random.default_rng(61)
    batches = []
    for i in range(len(pr)):
        fb = int(pr["face_begin"][i])
        polys = []
        for p in range(int(pr["n_poly"][i])):
            f0, f1 = fb + pr["face_off"][i][p], fb + pr["face_off"][i][p + 1]
            A, b = faces["a"][f0:f1].copy(), faces["b"][f0:f1].copy()
            extra = abi.FH_MAX_FACES_POLY - len(b) if (i % 3 == 0 and p == 0) else 3
            idx = rng.integers(0, len(b), size=extra)
            A2 = np.vstack([A, A[idx] * rng.uniform(0.5, 2.0, size=(extra, 1))])      # scaled copies: same half-spaces
            b2 = np.concatenate([b, b[idx] * (np.linalg.norm(A2[len(b):], axis=1) / np.linalg.norm(A[idx], axis=1)) + rng.uniform(0, 0.5, size=extra)])
            if i % 4 == 1:                      # harmless zero-normal row
                A2 = np.vstack([A2, np.zeros((1, 3))]); b2 = np.append(b2, 0.25)
            if i % 8 == 2 and p == 1:           # infeasible zero-normal row: polytope p can never be used
                A2 = np.vstack([A2, np.zeros((1, 3))]); b2 = np.append(b2, -0.25)
            polys.append((A2, b2))
        f, off = abi.pack_faces(polys)
        q = pr[i: i + 1].copy()
        q["face_begin"] = 0
        q["face_off"][0, : len(off)] = off
        q["face_off"][0, len(off):] = off[-1]
        batches.append((q, f))
    pr2, faces2 = corridor.concat(batches)
    got = ctx.solve_batch(pr2, faces2)
    ref = oracle.solve_batch(pr2, faces2)
    compare(got, ref)


def test_limits_and_params(oracle):
    """fh_set_params: node / iteration caps are reported, never hang; a looser feas_tol is honoured."""
    import torch  # noqa: F401
    from faster_amd import capi

    c = capi.Context(0)
    pr, faces, _ = corridor.whole_batch(64, seed=71, p_choices=(5, 6))
    par = abi.default_params()
    par["max_nodes"] = 2
    c.set_params(par)
    got = c.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces, params=par)
    assert set(np.unique(got["status"])) <= {abi.FH_ST_OPTIMAL, abi.FH_ST_NODE_LIMIT, abi.FH_ST_INFEASIBLE}
    assert (got["status"] == abi.FH_ST_NODE_LIMIT).any()
    assert np.all(got["solved"][got["status"] == abi.FH_ST_NODE_LIMIT] == 0)
    assert (ref["status"] == abi.FH_ST_NODE_LIMIT).any()
    par = abi.default_params()
    par["max_iters"] = 3
    c.set_params(par)
    got = c.solve_batch(pr, faces)
    assert (got["status"] == abi.FH_ST_ITER_LIMIT).any() and np.all(got["solved"][got["status"] == abi.FH_ST_ITER_LIMIT] == 0)
    par = abi.default_params()
    par["feas_tol"] = 1e-6      # Gurobi's FeasibilityTol
    c.set_params(par)
    got = c.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces, params=par)
    compare(got, ref, cost_rtol=1e-5, coeff_atol=1e-4)
    with pytest.raises(capi.FasterHipError):
        bad = abi.default_params()
        bad["feas_tol"] = 0.0
        c.set_params(bad)
    c.close()


def test_two_contexts_two_streams(oracle):
    """Handles are independent: two contexts on two HIP streams solving different batches concurrently."""
    import torch
    from faster_amd import capi

    c1, c2 = capi.Context(0), capi.Context(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c1.set_stream(s1.cuda_stream)
    c2.set_stream(s2.cuda_stream)
    b1 = corridor.whole_batch(512, seed=81)[:2]
    b2 = corridor.safe_batch(512, seed=82, n_seg=10, p_choices=(1, 2, 3))[:2]
    outs = []
    for rep in range(3):  # repeated launches on the same contexts (the work counter is never reset)
        d = []
        for c, (pr, faces) in ((c1, b1), (c2, b2)):
            n = len(pr)
            mf = int(pr["face_off"][np.arange(n), pr["n_poly"]].max())
            dp, df = _dev(pr), _dev(faces)
            dr = torch.zeros(n * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
            c.solve_batch_device(dp.data_ptr(), df.data_ptr(), n, int(pr["n_seg"].max()), mf, dr.data_ptr())
            d.append((dp, df, dr))
        c1.sync(); c2.sync()
        outs = [x[2].cpu().numpy().view(abi.result_dtype) for x in d]
    compare(outs[0], oracle.solve_batch(*b1))
    compare(outs[1], oracle.solve_batch(*b2))
    c1.close(); c2.close()


def test_sample_capacity_and_unsolved(ctx, oracle):
    """fh_sample_batch: counts report the full size even when max_samples truncates; unsolved problems give 0."""
    pr, faces, _ = corridor.whole_batch(16, seed=91)
    pr = pr.copy()
    pr["x0"][0, 3] = 9.0   # infeasible start
    res = ctx.solve_batch(pr, faces)
    states, counts = ctx.sample_batch(pr, res, 10)
    assert counts[0] == 0 and res["solved"][0] == 0
    for i in range(1, len(pr)):
        full = oracle.sample(pr[i], res[i])
        assert counts[i] == full.shape[0] > 10
        np.testing.assert_allclose(states[i]["pos"], full[:10]["pos"], atol=1e-11)


def test_fixed_binaries_config_c1(ctx, oracle, fixture_corridor, known_answers):
    """BASELINE config 1 "N=6, 3-polytope corridor, fixed binaries (pure QP)" and the pinned runners-up of KA-1:
    fh_problem.pin fixes b[t][p]; costs are the survey's independent values (tests/golden/known_answers.json)."""
    c = known_answers["cases"]["KA-1"]
    batches, expect = [], []
    for cost, assign in [(c["cost"], c["assign"])] + [tuple(x) for x in c["runners_up"]]:
        pr, faces = corridor.fixture_problem(fixture_corridor, c["N"], c["vaj"], 1, c["polys"], c["x0"], c["xf"], f_init=3.0, f_final=3.0)
        abi.set_pins(pr[0], assign)
        batches.append((pr, faces))
        expect.append(cost)
    c2 = known_answers["cases"]["KA-2"]   # C1 proper: N=6, 3 polytopes, monotone pattern fixed
    pr, faces = corridor.fixture_problem(fixture_corridor, 6, c2["vaj"], 1, c2["polys"], c2["x0"], c2["xf"])
    abi.set_pins(pr[0], c2["assign"])
    batches.append((pr, faces))
    pr, faces = corridor.fixture_problem(fixture_corridor, 6, c2["vaj"], 1, c2["polys"], c2["x0"], c2["xf"])
    abi.set_pins(pr[0], [0, 0, 1, 1, 2, 2])   # a pattern that is infeasible for every factor in the window? compare with the oracle
    batches.append((pr, faces))
    prs, fcs = corridor.concat(batches)
    got = ctx.solve_batch(prs, fcs)
    ref = oracle.solve_batch(prs, fcs)
    compare(got, ref)
    for i, cost in enumerate(expect):
        assert got["solved"][i] == 1 and got["cost"][i] == pytest.approx(cost, rel=2e-8)
        assert list(got["assign"][i][:10]) == list(([c["assign"]] + [x[1] for x in c["runners_up"]])[i])
    assert got["solved"][len(expect)] == 1 and got["factor"][len(expect)] == 3.0
    assert list(got["assign"][len(expect)][:6]) == c2["assign"]
    # random pins on synthetic corridors, including pins that exclude every assignment
    pr, faces, _ = corridor.whole_batch(128, seed=111, n_seg=8, p_choices=(2, 3, 4))
    rng = np.random.default_rng(111)
    for i in range(len(pr)):
        a = [int(rng.integers(0, pr["n_poly"][i])) if rng.random() < 0.3 else -1 for _ in range(8)]
        abi.set_pins(pr[i], a)
    pr["pin"][5, 0] = 0x9           # polytope index 8 >= n_poly: bad input
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)
    assert got["status"][5] == abi.FH_ST_BAD_INPUT
    assert 0 < got["solved"].mean() < 1


def test_work_cap_bounds_the_worst_case(oracle):
    """fh_params.max_work: a per-problem cap on active-set iterations (the reference sets no Gurobi TimeLimit)."""
    import torch  # noqa: F401
    from faster_amd import capi

    c = capi.Context(0)
    pr, faces, _ = corridor.whole_batch(256, seed=3, p_choices=(5, 6))
    par = abi.default_params()
    par["max_work"] = 60
    c.set_params(par)
    got = c.solve_batch(pr, faces)
    full = oracle.solve_batch(pr, faces)
    capped = got["status"] == abi.FH_ST_ITER_LIMIT
    assert capped.any() and np.all(got["solved"][capped] == 0)
    assert np.all(got["qp_iters"] <= 60 + par["max_iters"])       # the cap is checked between nodes
    fine = ~capped
    compare(got[fine], full[fine])                                  # problems below the cap are unaffected
    c.close()


def test_forest_corridors_config_c5(ctx, oracle):
    """BASELINE config 5 inputs: corridors from the voxel path search + ellipsoid decomposition front-end
    (faster_amd/host/corridor_frontend.hpp) in a random forest, N=15, <=8 polytopes."""
    from faster_amd import build as fb, frontend

    fb.build_frontend()
    pr, faces, info = frontend.forest_batch(384, seed=5, n_seg=15, max_poly=8)
    assert len(pr) > 300 and info["overflow"] == 0
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    ok = compare(got, ref)
    assert ok.mean() > 0.8
    check_assignment_valid(pr, faces, got)
    pr2, faces2, _ = frontend.forest_batch(384, seed=6, n_seg=10, max_poly=6, force_final=False)
    compare(ctx.solve_batch(pr2, faces2), oracle.solve_batch(pr2, faces2))


def test_gpu_decomposition_matches_host_frontend(ctx):
    """Next row N1 on the device: fh_decompose_batch against the host front-end (faster_amd/host/corridor_frontend.hpp) on random
    scenes and on forest paths; rows compared bit for bit and in order (both sides: correctly rounded operations only — no libm
    trigonometry, no fused multiply-adds — and lowest-index tie rules)."""
    from faster_amd import build as fb, frontend

    fb.build_frontend()
    key = lambda M: M[np.lexsort(np.round(M, 7).T[::-1])]
    rng = np.random.default_rng(7)
    total = 0
    for scene in range(6):
        if scene < 4:
            path = np.cumsum(np.vstack([rng.uniform(-3, 3, 3) * [1, 1, 0] + [0, 0, 1.2], rng.uniform(0.8, 2.5, (4, 1)) * (rng.normal(size=(4, 3)) * [1, 1, 0.2])]), axis=0)
            path[:, 2] = np.clip(path[:, 2], 0.6, 2.4)
            cloud = rng.uniform(path.min(0) - 2.5, path.max(0) + 2.5, size=(700, 3))
            keep = np.ones(len(cloud), bool)
            for a, b in zip(path[:-1], path[1:]):
                t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
                keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > 0.45
            cloud = cloud[keep]
        else:
            cloud, _ = frontend.forest_cloud(scene)
            path = frontend.plan(cloud, (110, 110, 15), 0.2, np.array([10.0, 10.0, 1.5]), 0.0, 3.0, 0.3, np.array([1.5, 1.2, 1.0]), np.array([17.5, 18.0, 1.4]))
            assert path is not None
        segs = np.hstack([path[:-1], path[1:]])
        faces, counts = ctx.decompose_batch(cloud, segs, drone_radius=0.05, z_ground=0.0, max_faces=64)
        ref, _ = frontend.decompose(path, cloud, drone_radius=0.05, z_ground=0.0)
        for i, (A, b) in enumerate(ref):
            assert counts[i] == len(b), (scene, i, counts[i], len(b))
            got = np.column_stack([faces["a"][i, :counts[i]], faces["b"][i, :counts[i]]])
            assert np.array_equal(got, np.column_stack([A, b])), (scene, i)   # same rows, same order, bit for bit
            total += 1
    assert total >= 20
    # overflow reporting: too few rows allowed
    faces, counts = ctx.decompose_batch(cloud, segs[:2], max_faces=8)
    assert np.all(counts == -1)


def test_concurrent_factor_search_is_the_sequential_rule(ctx, oracle):
    """the line search run `width` factors at a time returns the sequential first-feasible result bit
    for bit (factor_that_worked_ semantics of solverGurobi.cpp:445-467), incl. unsolved, empty-window and bad records."""
    whole, faces, _ = corridor.whole_batch(96, seed=21, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    safe, sfaces, _ = corridor.safe_batch(64, seed=22)
    pr, fc = corridor.concat([(whole, faces), (safe, sfaces)])
    pr = pr.copy()
    pr["f_inc"][:40] = 0.5                      # 19 factors in [1, 10]
    pr["f_final"][40:48] = 1.5                  # short windows: mostly unsolved
    pr["f_init"][48:50] = 11.0                  # empty window
    pr["n_seg"][50] = 0                         # bad record
    pr["f_inc"][51] = 0.0                       # bad window
    seq = ctx.solve_batch(pr, fc)
    assert (seq["solved"] == 1).sum() > 60 and (seq["solved"] == 0).sum() > 8 and seq["trials"].max() >= 4
    compare(seq, oracle.solve_batch(pr, fc))   # the sequential rule itself is the oracle's (edge-case windows included)
    # the work counters (nodes, qp_iters, kflops) depend on who explored what when subtrees are shared between wavefronts:
    # everything genNewTraj() leaves behind is compared in the default mode, the counters as well with one wavefront per problem
    results = [n for n in abi.result_dtype.names if n not in ("nodes", "qp_iters", "kflops")]
    for width in (2, 3, 10, 64):
        got = ctx.solve_batch_speculative(pr, fc, width)
        for name in results:
            assert np.array_equal(got[name], seq[name]), (width, name)
    assert np.array_equal(ctx.solve_batch_speculative(pr, fc, 1)["coeff"], seq["coeff"])
    solo = capi.Context(0)
    par = abi.default_params()
    par["share"] = 0
    solo.set_params(par)
    seq1 = solo.solve_batch(pr, fc)
    for name in results:
        assert np.array_equal(seq1[name], seq[name]), name
    for width in (2, 10):
        got = solo.solve_batch_speculative(pr, fc, width)
        for name in abi.result_dtype.names:
            if name == "kflops":  # thousands of flops, rounded down per launch: per trial here, per problem there
                assert np.all(np.abs(got[name].astype(np.int64) - seq1[name]) <= np.maximum(seq1["trials"], 1)), width
            else:
                assert np.array_equal(got[name], seq1[name]), (width, name)
    solo.close()


def test_gpu_decomposition_edge_cases(ctx):
    """Empty cloud (box + ground plane only), a vertical segment (degenerate horizontal direction, line_segment.h:62-66), a cloud
    denser than the LDS list (points spill to the HBM workspace) and a batch larger than the resident grid."""
    from faster_amd import build as fb, frontend

    fb.build_frontend()
    key = lambda M: M[np.lexsort(np.round(M, 7).T[::-1])]

    def check(cloud, path, **kw):
        segs = np.hstack([path[:-1], path[1:]])
        faces, counts = ctx.decompose_batch(cloud, segs, max_faces=96, **kw)
        ref, _ = frontend.decompose(path, cloud, **kw)
        for i, (A, b) in enumerate(ref):
            assert counts[i] == len(b), (i, counts[i], len(b))
            got = np.column_stack([faces["a"][i, :counts[i]], faces["b"][i, :counts[i]]])
            assert np.array_equal(got, np.column_stack([A, b])), (scene, i)   # same rows, same order, bit for bit
        return counts

    path = np.array([[0.0, 0.0, 1.0], [1.5, 0.5, 1.2], [1.5, 0.5, 2.4], [3.0, 0.0, 2.0]])  # middle leg is vertical
    c = check(np.zeros((0, 3)), path)
    assert np.all(c == 7)
    rng = np.random.default_rng(11)
    dense = rng.uniform([-2.5, -2.5, 0.0], [5.5, 3.0, 3.5], size=(9000, 3))  # > 1024 points in every local box
    keep = np.ones(len(dense), bool)
    for a, b in zip(path[:-1], path[1:]):
        t = np.clip(((dense - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
        keep &= np.linalg.norm(dense - (a + t[:, None] * (b - a)), axis=1) > 0.4
    dense = dense[keep]
    check(dense, path, drone_radius=0.1)
    # the three homes of a segment's point list: coordinates in LDS (<= 256 points), ids in LDS (<= 1536), ids in the workgroup's
    # HBM workspace (the sweep is repeated into it) — one cloud per regime, the same rows as the host every time
    for size, lo, hi in ((600, 1, 256), (9000, 257, 1536), (45000, 1537, 16384)):
        cl = rng.uniform([-2.5, -2.5, 0.0], [5.5, 3.0, 3.5], size=(size, 3))
        keep = np.ones(len(cl), bool)
        for a, b in zip(path[:-1], path[1:]):
            t = np.clip(((cl - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
            keep &= np.linalg.norm(cl - (a + t[:, None] * (b - a)), axis=1) > 0.4
        cl = cl[keep]
        inside = []
        for a, b in zip(path[:-1], path[1:]):  # points of the local box (2 x 2 x 1 m around the leg), roughly: which regime the leg is in
            d = (b - a) / np.linalg.norm(b - a)
            rel = cl - a
            along = rel @ d
            perp = np.linalg.norm(rel - np.outer(along, d), axis=1)
            inside.append(int(((along > -2.0) & (along < np.linalg.norm(b - a) + 2.0) & (perp < 1.0)).sum()))
        assert any(lo <= k <= hi for k in inside), (size, inside)
        check(cl, path, drone_radius=0.1)
    # many segments (more than resident workgroups): every copy of a segment gives the same polytope
    segs = np.tile(np.hstack([path[:-1], path[1:]]), (1500, 1))
    faces, counts = ctx.decompose_batch(dense[::8], segs, max_faces=96)
    for k in range(3):
        assert np.all(counts[k::3] == counts[k])
        assert np.array_equal(faces["b"][k::3], np.broadcast_to(faces["b"][k], faces["b"][k::3].shape))


def test_fused_pair_kernel_equals_three_launches(ctx):
    """fh_solve_pairs_device (one launch, a wavefront per pair: whole -> hand-off -> safe) against the three-launch pipeline:
    same device functions, bit-identical whole results, safe problems, safe faces and safe results."""
    import torch

    B, N = 1536, 10
    whole, faces, _ = corridor.whole_batch(B, seed=31, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    whole = whole.copy()
    whole["f_final"][:16] = 1.0  # some whole problems without a solution: the pair ends there (safe n_seg = 0)
    safe_t = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    dev = "cuda:0"

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    d_whole, d_faces = to_dev(whole), to_dev(faces)
    outs = []
    for fused in (False, True):
        d_safe, d_sf = to_dev(safe_t), torch.zeros_like(d_faces)
        d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros_like(d_wr)
        if fused:
            ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(),
                                   d_sf.data_ptr(), d_sr.data_ptr())
        else:
            ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_wr.data_ptr())
            ctx.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
            ctx.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, mf, d_sr.data_ptr())
        ctx.sync()
        outs.append((d_wr.cpu().numpy().copy(), d_safe.cpu().numpy().copy(), d_sf.cpu().numpy().copy(), d_sr.cpu().numpy().copy()))
    for a, b, name in zip(outs[0], outs[1], ("whole results", "safe problems", "safe faces", "safe results")):
        if name.endswith("results"):  # (the work counters depend on which wavefronts shared a tree)
            ra, rb = a.view(abi.result_dtype), b.view(abi.result_dtype)
            for f in abi.result_dtype.names:
                if f not in ("nodes", "qp_iters", "kflops"):
                    assert np.array_equal(ra[f], rb[f]), (name, f)
        else:
            assert np.array_equal(a, b), name
    sres = outs[1][3].view(abi.result_dtype)
    assert (sres["solved"] == 1).sum() > B // 2 and (outs[1][1].view(abi.problem_dtype)["n_seg"][:16] == 0).all()


def test_full_size_c4_properties(ctx):
    """BASELINE config C4 at its full size (32768 whole+safe pairs) through size-independent properties of the model
    (solverGurobi.cpp:332-407, :217-290): boundary conditions, C2 continuity, box limits at the segment starts, every control
    point inside the polytope the result assigns its segment to, the reported cost, the factor window, and the hand-off."""
    import torch

    B, N = 32768, 10
    whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(corridor.safe_templates(whole))
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_sr = torch.zeros_like(d_wr)
    ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(),
                           d_sf.data_ptr(), d_sr.data_ptr())
    ctx.sync()
    wres, sres = d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype)
    safe, sfaces = d_safe.cpu().numpy().view(abi.problem_dtype), d_sf.cpu().numpy().view(abi.face_dtype)
    assert wres["solved"].mean() > 0.99 and 0.6 < sres["solved"].mean() < 0.95

    def check(pr, fc, rs, force_final):
        ok = rs["solved"] == 1
        p, r = pr[ok], rs[ok]
        n = len(p)
        h = r["dt"][:, None, None]
        c = r["coeff"][:, :N, :].reshape(n, N, 4, 3)                     # [problem, segment, (a, b, c, d), axis]
        a3, b2, c1, d0 = c[:, :, 0], c[:, :, 1], c[:, :, 2], c[:, :, 3]
        end = a3 * h**3 + b2 * h**2 + c1 * h + d0
        vend = 3 * a3 * h**2 + 2 * b2 * h + c1
        aend = 6 * a3 * h + 2 * b2
        x0 = p["x0"].reshape(n, 3, 3)
        scale = 1.0 + np.abs(p["x0"]).max()
        assert np.abs(d0[:, 0] - x0[:, 0]).max() < 1e-9 * scale and np.abs(c1[:, 0] - x0[:, 1]).max() < 1e-9 and np.abs(2 * b2[:, 0] - x0[:, 2]).max() < 1e-9
        assert np.abs(end[:, :-1] - d0[:, 1:]).max() < 1e-8 and np.abs(vend[:, :-1] - c1[:, 1:]).max() < 1e-8
        assert np.abs(aend[:, :-1] - 2 * b2[:, 1:]).max() < 1e-7          # C2 continuity (setDynamicConstraints)
        xf = p["xf"].reshape(n, 3, 3)
        assert np.abs(vend[:, -1] - xf[:, 1]).max() < 1e-7 and np.abs(aend[:, -1] - xf[:, 2]).max() < 1e-7
        if force_final:
            assert np.abs(end[:, -1] - xf[:, 0]).max() < 1e-7
        lim = 1e-7
        assert (np.abs(c1) <= p["v_max"][:, None, None] + lim).all() and (np.abs(2 * b2) <= p["a_max"][:, None, None] + lim).all()
        assert (np.abs(6 * a3) <= p["j_max"][:, None, None] + lim).all()   # boxes at the segment starts only (setMaxConstraints)
        cost = ((6 * a3) ** 2).sum(axis=(1, 2))
        np.testing.assert_allclose(r["cost"], cost, rtol=1e-9, atol=1e-9)
        assert (r["factor"] >= p["f_init"]).all() and (r["factor"] <= p["f_final"]).all() and (r["trials"] >= 1).all()
        # control points in the assigned polytope: worst violation over all (problem, segment, control point, face)
        cps = np.stack([d0, d0 + c1 * h / 3, d0 + 2 * c1 * h / 3 + b2 * h * h / 3, end], axis=2)   # [n, N, 4, 3]
        q = r["assign"][:, :N].astype(np.int64)
        assert (q >= 0).all() and (q < p["n_poly"][:, None]).all()
        f0 = p["face_begin"][:, None] + np.take_along_axis(p["face_off"], q, axis=1)
        f1 = p["face_begin"][:, None] + np.take_along_axis(p["face_off"], q + 1, axis=1)
        worst = -np.inf
        for k in range(int((f1 - f0).max())):
            live = (f0 + k) < f1
            idx = np.where(live, f0 + k, 0)
            val = (fc["a"][idx][:, :, None, :] * cps).sum(axis=3) - fc["b"][idx][:, :, None]
            worst = max(worst, float(np.where(live[:, :, None], val, -np.inf).max()))
        assert worst <= 1e-6, worst
        return ok

    okw = check(whole, faces, wres, True)
    oks = check(safe, sfaces, sres, False)
    assert (safe["n_seg"][~okw] == 0).all() and (sres["solved"][~okw] == 0).all()   # no whole trajectory: the pair ends there
    assert (oks <= okw).all()


def test_full_size_translation_invariance(ctx):
    """32768 whole problems and the same problems translated by a fixed vector (positions of x0/xf shifted, b += a.shift): same
    feasibility, trials and factor; same cost and same polynomial but for the constant term (the model has no absolute frame)."""
    B, N = 32768, 10
    whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    shift = np.array([3.25, -1.5, 0.75])
    w2, f2 = whole.copy(), faces.copy()
    w2["x0"][:, :3] += shift
    w2["xf"][:, :3] += shift
    f2["b"] += f2["a"] @ shift
    r1, r2 = ctx.solve_batch(whole, faces), ctx.solve_batch(w2, f2)
    same = (r1["solved"] == r2["solved"]) & (r1["trials"] == r2["trials"]) & (r1["factor"] == r2["factor"])
    assert same.mean() > 0.9995            # (a feasibility margin below the shift's rounding may flip: none expected, a few tolerated)
    ok = same & (r1["solved"] == 1)
    np.testing.assert_allclose(r2["cost"][ok], r1["cost"][ok], rtol=1e-7, atol=1e-9)
    c1, c2 = r1["coeff"][ok][:, :N].reshape(-1, N, 4, 3), r2["coeff"][ok][:, :N].reshape(-1, N, 4, 3)
    np.testing.assert_allclose(c2[:, :, :3], c1[:, :, :3], rtol=0, atol=1e-6)
    np.testing.assert_allclose(c2[:, :, 3] - shift, c1[:, :, 3], rtol=0, atol=1e-6)
