"""GPU tests added in round 5 (all through the C ABI): the time allocation on its own entry point at a scale where the single-precision
starting points of its root finder would show (fh_dt_initial_batch), trials refuted at y = 0 by the jerk box, rule mode 2 inside the
fused pair kernel."""
import numpy as np
import pytest
import torch  # noqa: F401  (before libfasterhip.so is loaded: one HIP runtime per process)

from faster_amd import abi, capi, corridor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = capi.Context(0)
    yield c
    c.close()


def _dt_problems(n, rng, kind):
    pr = np.zeros(n, dtype=abi.problem_dtype)
    pr["n_seg"] = rng.choice([3, 6, 10, 15, 16], n)
    pr["dc"] = 0.01
    pr["v_max"] = rng.choice([1.5, 3.0, 5.0, 20.0], n)
    pr["a_max"] = rng.choice([2.0, 3.0, 5.0, 9.0], n)
    pr["j_max"] = rng.choice([1.0, 5.0, 8.0, 30.0], n)
    pr["f_init"], pr["f_final"], pr["f_inc"] = 1.0, 10.0, 1.0
    x0 = np.zeros((n, 9))
    xf = np.zeros((n, 9))
    x0[:, :3] = rng.uniform(-20, 20, (n, 3))
    xf[:, :3] = x0[:, :3] + rng.uniform(-12, 12, (n, 3)) * rng.choice([1.0, 1.0, 0.1, 1e-3], (n, 1))
    x0[:, 3:6] = rng.uniform(-1, 1, (n, 3)) * pr["v_max"][:, None]
    x0[:, 6:9] = rng.uniform(-1, 1, (n, 3)) * pr["a_max"][:, None]
    if kind == "rest":          # FASTER's usual start of a mission and every goal: zero velocity and acceleration
        x0[:, 3:] = 0.0
    elif kind == "axis":        # nothing to do on some axes (dx = 0 exactly), or only a velocity / only an acceleration
        m = rng.random((n, 3)) < 0.5
        xf[:, :3] = np.where(m, x0[:, :3], xf[:, :3])
        x0[:, 3:6] *= rng.random((n, 3)) < 0.5
        x0[:, 6:9] *= rng.random((n, 3)) < 0.5
    elif kind == "scale":       # kilometres and millimetres
        s = rng.choice([1e3, 1e-3, 1e-6], (n, 1))
        xf[:, :3] = x0[:, :3] + (xf[:, :3] - x0[:, :3]) * s
    elif kind == "close":       # cubics with two roots at a relative distance eps (and a third one elsewhere): the constant-jerk cubic
        j = pr["j_max"].astype(np.float32).astype(np.float64) / 6.0   # (j / 6)(t - t1)(t - t2)(t - t3) with dx > 0
        t1 = rng.uniform(0.2, 4.0, (n, 3))
        eps = 10.0 ** rng.uniform(-9, -2, (n, 3))
        t2 = t1 * (1.0 + eps)
        t3 = rng.uniform(0.2, 6.0, (n, 3)) * rng.choice([1.0, -1.0], (n, 3))
        neg = t3 < 0                                 # dx = (j / 6) t1 t2 t3 must be positive: flip the pair with the third root
        t1, t2 = np.where(neg, -t1, t1), np.where(neg, -t2, t2)
        x0[:, 6:9] = -2.0 * j[:, None] * (t1 + t2 + t3)
        x0[:, 3:6] = j[:, None] * (t1 * t2 + t1 * t3 + t2 * t3)
        xf[:, :3] = x0[:, :3] + j[:, None] * (t1 * t2 * t3)
    pr["x0"], pr["xf"] = x0, xf
    return pr


def test_time_allocation_alone_on_two_million_problems(ctx, oracle):
    """Row a5 — getDTInitial (solverGurobi.cpp:659-759) through its own entry point, where a difference could not hide behind a solve:
    the device finds the roots from single-precision starting points polished in double (fh_solve.hip.hpp: dt_initial) and falls back
    to the oracle's closed forms when two roots are close; the value is stored as a float by the reference.  Bit-identical to the
    oracle on 2 M problems of four kinds (generic states, at rest, idle axes, kilometres / micrometres).  Cubics BUILT with two roots
    at relative distances 1e-9 .. 1e-2 — where the root itself is only defined to ~1e-8 by either library's cbrt / acos — must agree to
    the float the reference stores (2 ulp), and nearly all of them exactly.
    [r6] dx = 0 exactly on an axis puts a root of that axis' cubic AT zero; the general closed form returns it as +1e-17, 0 or -1e-17 by
    the last bit of cbrt / acos / cos (round 5 counted the disagreements between the device library and glibc: 0.025 %; Eigen's
    companion-matrix eigenvalues would have their own).  Device and oracle now share a stated convention (fasterhip.h, faster_oracle.c:
    real_roots_cubic): the cubic factors as t (c3 t^2 + c2 t + c1), the zero root is exactly zero — MinPositiveElement drops it — and
    the others come from the quadratic's closed form.  That class is bit-identical too."""
    rng = np.random.default_rng(505)
    n = 1 << 19
    for kind in ("generic", "rest", "axis", "scale"):
        pr = _dt_problems(n, rng, kind)
        got, ref = ctx.dt_initial_batch(pr), oracle.dt_initial_batch(pr)
        at_zero = np.any(pr["xf"][:, :3] == pr["x0"][:, :3], axis=1)
        bad = np.nonzero(got != ref)[0]
        assert len(bad) == 0, (kind, len(bad), int(at_zero[bad].sum()), got[bad[:4]], ref[bad[:4]], pr["x0"][bad[:2]], pr["xf"][bad[:2]])
        if kind == "axis":
            assert at_zero.sum() > n // 2  # (the class the convention is about is really exercised)
        assert (ref > 0).mean() > 0.8
    pr = _dt_problems(n, rng, "close")
    got, ref = ctx.dt_initial_batch(pr), oracle.dt_initial_batch(pr)
    np.testing.assert_allclose(got, ref, rtol=2.5e-7, atol=0)
    assert (got == ref).mean() > 0.999, (got != ref).sum()
    # and inside a solve: dt = factor * max(dt_initial, 2 dc) of the first trial, bit for bit
    w, faces, _ = corridor.whole_batch(512, seed=11, n_seg=10, p_choices=(2, 3))
    res = ctx.solve_batch(w, faces)
    d0 = np.maximum(oracle.dt_initial_batch(w), 2 * w["dc"])
    ok = res["solved"] == 1
    assert ok.sum() > 400 and np.array_equal(res["dt"][ok], res["factor"][ok] * d0[ok])


def test_unreachable_goals_on_a_big_map_with_the_default_records():
    """with the records chosen by the size of the map (fh_map_set_records -1) a 1.4 M-cell map searches with a hashed table of
    131072 slots per wavefront, and a query that reaches more cells than the table holds ends at that limit (n_points -2).  A goal sealed
    inside a shell of obstacle points is such a query: the jump point search has to exhaust the forest.  The host-pointer entry point
    runs those queries again with one record per cell, so that its answers are those of fh_map_set_records(0) — "no path" (0), as jps3d
    answers (graph_search.cpp:219-221: the open set runs empty) — and every other query is untouched; the asynchronous device-pointer
    entry point reports -2 for them."""
    import torch

    from faster_amd import frontend

    res, infl, zmax = 0.2, 0.3, 20.0                   # a forest 20 m tall: 110 x 110 x 100 cells, and three dimensions to exhaust
    cloud, cells, center, starts, goals, rng = frontend.forest_queries(48, 29, size=(20.0, 20.0, zmax), res=res, inflation=infl, return_rng=True)
    sealed = np.arange(0, 48, 8)                        # six goals inside a closed shell (radius 1.1 m, far outside the cube freed around a goal)
    shell = []
    for i in sealed:
        goals[i] = [10.0 + 0.5 * (i // 8), 10.0, 10.0]
        v = rng.normal(size=(6000, 3))
        shell.append(goals[i] + 1.1 * v / np.linalg.norm(v, axis=1, keepdims=True))
    cloud = np.concatenate([cloud] + shell)
    far = np.linalg.norm(starts[:, None, :] - goals[sealed][None, :, :], axis=2).min(axis=1) > 2.5
    starts[~far] = [1.0, 1.0, 15.0]                     # no start inside a shell
    m = capi.Map(0)
    try:
        m.set_search("jps")
        m.read(cloud, cells, res, center, 0.0, zmax, infl)
        assert np.prod(m.dims()[0]) > 1_000_000
        m.set_records(0)
        want = m.plan_batch(starts, goals)              # one record per cell: no limit on the cells a query reaches
        assert np.all(want[1][sealed] == 0) and (want[1] >= 2).sum() >= 36, want[1]
        dense_bytes = m.workspace_bytes()
        m.set_records(-1)                               # the default
        d_s, d_g = torch.from_numpy(starts).cuda(), torch.from_numpy(goals).cuda()
        d_p = torch.zeros((48, 64, 3), dtype=torch.float64, device="cuda:0")
        d_n = torch.zeros(48, dtype=torch.int32, device="cuda:0")
        d_e = torch.zeros(48, dtype=torch.int64, device="cuda:0")
        m.plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), 48, 64, d_p.data_ptr(), d_n.data_ptr(), d_e.data_ptr())
        m.sync()
        dev_n = d_n.cpu().numpy()
        hashed = m.workspace_bytes() < 0.8 * dense_bytes
        if not (hashed and (dev_n == -2).any()):
            pytest.skip("the default did not choose a hashed table that these queries overflow (workspace %.1f GB, n_points of the sealed goals %s)"
                        % (m.workspace_bytes() / 1e9, dev_n[sealed]))
        assert set(np.nonzero(dev_n == -2)[0]) <= set(sealed)
        got = m.plan_batch(starts, goals)               # host pointers: the queries at the limit are run again with per-cell records
        assert np.array_equal(got[1], want[1]), (got[1], want[1])
        assert np.array_equal(got[2], want[2])          # the same nodes popped
        for i in np.nonzero(want[1] > 0)[0]:
            assert np.array_equal(got[0][i, :want[1][i]], want[0][i, :want[1][i]]), i
    finally:
        m.close()


def test_pool_with_unknown_space_as_an_input_equals_one_context(ctx):
    """fh_pool_set_unknown_grid + fh_pool_set_pair_rule(mode 2): one batch of pairs over a pool (the device named three times: three
    contexts, three copies of the unknown voxels, three shards) equals the fused pair kernel of ONE context with the same grid, record for
    record — whole and safe results; and the pool refuses mode 2 until it has a grid."""
    rng = np.random.default_rng(77)
    B = 301
    whole, faces, _ = corridor.whole_batch(B, seed=21, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    tmpl = corridor.safe_templates(whole)
    res, dims = 0.25, (96, 96, 16)
    origin = np.array([whole["x0"][:, 0].min() - 2.0, whole["x0"][:, 1].min() - 2.0, -0.5])
    flags = (rng.random(dims[::-1]) < 0.0008).astype(np.uint8)          # [nz][ny][nx]: a few scattered unknown voxels
    rule = dict(mode=2, drone_radius=0.3, delta_h=1.0, delta_a=0.5)
    fields = [n for n in abi.result_dtype.names if n not in ("nodes", "qp_iters", "kflops")]
    # one context: the fused pair kernel
    d_flags = torch.from_numpy(flags.reshape(-1).copy()).cuda()
    ctx.set_pair_rule(**rule)
    ctx.set_unknown_grid_device(d_flags.data_ptr(), origin, res, dims)
    ctx.set_pair_margin(0.0)
    try:
        mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()
        d_w, d_f, d_s = to_dev(whole), to_dev(faces), to_dev(tmpl)
        d_sf = torch.zeros_like(d_f)
        d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
        d_sr = torch.zeros_like(d_wr)
        ctx.solve_pairs_device(d_w.data_ptr(), d_f.data_ptr(), B, 10, mf, 0.5, 0.0, 3, d_wr.data_ptr(), d_s.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync()
        assert ctx.last_launch()[1] == "fh::solve_kernel<10, true, 2, true>"
        w1, s1 = d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype)
        safe1 = d_s.cpu().numpy().view(abi.problem_dtype)
    finally:
        ctx.set_pair_rule(mode=0)
        ctx.set_pair_margin(-1.0)
        ctx.set_unknown_grid_device(None)
    need = safe1["n_seg"] > 0
    assert 0.05 * B < need.sum() < B and w1["solved"].sum() > 0.9 * B       # some trajectories come near an unknown voxel, some do not
    # the pool
    pool = capi.Pool([0, 0, 0])
    try:
        pool.set_pair_margin(0.0)
        pool.set_pair_rule(**rule)
        with pytest.raises(capi.FasterHipError):
            pool.solve_pairs(whole, faces, tmpl, 0.5, 0.0, 3)              # mode 2 without the unknown voxels
        pool.set_unknown_grid(flags, origin, res, dims)
        w3, s3 = pool.solve_pairs(whole, faces, tmpl, 0.5, 0.0, 3)
        for f in fields:
            assert np.array_equal(w3[f], w1[f]), f
            assert np.array_equal(s3[f][need], s1[f][need]), ("safe", f)
        assert not np.any(s3["solved"][~need])
    finally:
        pool.close()
