"""GPU tests added in round 3 (all through the C ABI): the kernel instantiations and batch sizes that the bench measures but the
earlier tests did not compare with the oracle — fused whole+safe pairs at N = 15 (`solve_kernel<15, true>`, the C5 kernel),
N = 16 and N = 6, on synthetic AND forest corridors from the device front-end; BASELINE config C3 at its full 4096; a 4096-pair
subsample of the full 65536-pair C5 batch; and the independent models (SciPy on the reference's unreduced 12N-coefficient
model) applied to 64 GPU results at N = 10."""
import numpy as np
import pytest

from faster_amd import abi, capi, corridor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    import torch  # noqa: F401  (torch before the HIP library: one HIP runtime in the process)


@pytest.fixture(scope="module")
def ctx():
    import torch  # noqa: F401  (torch first: one HIP runtime in the process)

    c = capi.Context(0)
    yield c
    c.close()


def compare(got, ref, cost_rtol=1e-7, coeff_atol=1e-6):
    assert np.array_equal(got["solved"], ref["solved"]), np.nonzero(got["solved"] != ref["solved"])
    assert np.array_equal(got["trials"], ref["trials"]), np.nonzero(got["trials"] != ref["trials"])
    assert np.array_equal(got["factor"], ref["factor"]) and np.array_equal(got["dt"], ref["dt"])
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["solved"] == 1
    np.testing.assert_allclose(got["cost"][ok], ref["cost"][ok], rtol=cost_rtol, atol=1e-9)
    np.testing.assert_allclose(got["coeff"][ok], ref["coeff"][ok], rtol=0, atol=coeff_atol)
    return ok


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def fused_pairs(ctx, whole, faces, tmpl, max_seg, margin):
    """One fused launch (whole -> hand-off -> safe per pair); returns host views of everything the device wrote."""
    import torch

    B = len(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    ctx.set_pair_margin(margin)
    d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(tmpl)
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_sr = torch.zeros_like(d_wr)
    ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, max_seg, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(),
                           d_sf.data_ptr(), d_sr.data_ptr())
    ctx.sync()
    ctx.set_pair_margin(-1.0)
    return (d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype),
            d_safe.cpu().numpy().view(abi.problem_dtype), d_sf.cpu().numpy().view(abi.face_dtype))


def check_pairs_against_oracle(ctx, oracle, whole, faces, max_seg, idx, margin=0.05, n_safe=None):
    """The pairs `idx` of a fused launch over the whole batch: whole results, hand-off records (against oracle/pair_glue.py driven by
    the ORACLE's whole results) and the safe results (the device-written safe problems solved by the oracle)."""
    from oracle import pair_glue

    tmpl = corridor.safe_templates(whole)
    wres, sres, safe, sfaces = fused_pairs(ctx, whole, faces, tmpl, max_seg, margin)
    wref = oracle.solve_batch(whole[idx], faces)
    compare(wres[idx], wref)
    safe_ref, _ = pair_glue.glue(whole[idx], wref, faces, tmpl[idx], 0.5, 0.2, 3, r_margin=margin)
    assert np.array_equal(safe["n_seg"][idx], safe_ref["n_seg"]) and np.array_equal(safe["n_poly"][idx], safe_ref["n_poly"])
    np.testing.assert_allclose(safe["x0"][idx], safe_ref["x0"], rtol=0, atol=1e-9)
    live = np.nonzero(safe_ref["n_seg"] > 0)[0]
    if n_safe is not None:
        live = live[:n_safe]
    sref = oracle.solve_batch(safe[idx][live], sfaces)
    oks = compare(sres[idx][live], sref)
    return wref, oks


@pytest.mark.parametrize("n_seg,p_choices,B", [(15, (4, 5, 6, 7, 8), 768), (16, (6, 7, 8), 192), (6, (1, 2, 3), 1024), (12, (3, 5), 256)])
def test_fused_pairs_against_oracle_synthetic(ctx, oracle, n_seg, p_choices, B):
    """solve_kernel<15, true> (the C5 kernel), <16, true>, <6, true> and a size in between: every pair against the oracle."""
    whole, faces, _ = corridor.whole_batch(B, seed=300 + n_seg, n_seg=n_seg, p_choices=p_choices)
    wref, oks = check_pairs_against_oracle(ctx, oracle, whole, faces, n_seg, np.arange(B))
    assert wref["solved"].mean() > 0.9 and oks.mean() > 0.4


def test_fused_pairs_against_oracle_forest_corridors(ctx, oracle):
    """Fused pairs at N = 15 on corridors produced by the DEVICE front-end (fh_map_* + fh_corridor_batch_device) in a random forest
    (BASELINE config 5's input distribution), and at N = 10 on the same kind of corridors with at most 6 polytopes."""
    from faster_amd import frontend

    vmap = capi.Map(0)
    try:
        for n_seg, max_poly, seed in ((15, 8, 51), (10, 6, 52)):
            whole, faces, info = frontend.forest_batch(640, seed=seed, n_seg=n_seg, max_poly=max_poly, front="device", ctx=ctx, vmap=vmap)
            assert len(whole) > 500 and info["no_path"] < 100
            wref, oks = check_pairs_against_oracle(ctx, oracle, whole, faces, n_seg, np.arange(len(whole)))
            assert wref["solved"].mean() > 0.8
    finally:
        vmap.close()


def test_config_c3_full_size_against_oracle(ctx, oracle):
    """BASELINE config C3 at its full size: 4096 whole-trajectory MIQPs, N = 10, <= 4 polytopes, every problem against the oracle."""
    pr, faces, _ = corridor.whole_batch(4096, seed=2, n_seg=10, p_choices=(2, 3, 4))
    got = ctx.solve_batch(pr, faces)
    ok = compare(got, oracle.solve_batch(pr, faces))
    assert ok.mean() > 0.95


def test_control_points_of_c3_and_c4_equal_the_oracles(ctx, oracle):
    """The outputs BASELINE names — control points, cost, feasibility flag — as such: getCP0..getCP3 (solverGurobi.cpp:833-862) of every
    segment of every solved problem of C3 (all 4096) and of a C4 subsample (whole and safe problems of 2048 pairs), the product's
    fh_control_points (the reference's literal expressions on the GPU's coefficients) against the oracle's (its jerk-space form on ITS
    coefficients), 1e-6 absolute — and each of them inside the polytope the result assigns its segment to."""
    from oracle import pair_glue

    N = 10
    pr, faces, _ = corridor.whole_batch(4096, seed=2, n_seg=N, p_choices=(2, 3, 4))
    whole, wfaces, _ = corridor.whole_batch(2048, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    wres = ctx.solve_batch(whole, wfaces)
    safe, sfaces = pair_glue.glue(whole, oracle.solve_batch(whole, wfaces), wfaces, corridor.safe_templates(whole), 0.5, 0.2, 3, r_margin=0.05)
    live = safe["n_seg"] > 0
    for name, (p, f, got) in {"C3": (pr, faces, None), "C4 whole": (whole, wfaces, wres), "C4 safe": (safe[live], sfaces, None)}.items():
        got = ctx.solve_batch(p, f) if got is None else got
        ref = oracle.solve_batch(p, f)
        ok = compare(got, ref)
        assert ok.sum() > 0.5 * len(p), name
        cg, co = capi.control_points(got, N), oracle.control_points(ref, N)
        assert cg.shape == (len(p), N, 4, 3) and not np.any(cg[~ok]) and not np.any(co[~ok])
        np.testing.assert_allclose(cg[ok], co[ok], rtol=0, atol=1e-6, err_msg=name)
        # cost and flag beside them: compare() above (flag exact, cost 1e-7 relative).  Inside the assigned polytope (:254-288):
        for i in np.nonzero(ok)[0][:512]:
            for t in range(N):
                q = int(got["assign"][i][t])
                f0, f1 = p[i]["face_begin"] + p[i]["face_off"][q], p[i]["face_begin"] + p[i]["face_off"][q + 1]
                assert np.max(f["a"][f0:f1] @ cg[i, t].T - f["b"][f0:f1, None]) <= 1e-6, (name, i, t)


def test_config_c5_full_size_subsample_against_oracle(ctx, oracle):
    """BASELINE config C5 at its full size: 65536 start/goal pairs in one random forest, corridors from the device front-end, N = 15,
    <= 8 polytopes, ONE fused launch over all of them (bench.py --workload c5); a random 4096-pair subsample — whole results,
    hand-off records, and the safe results of its first 2048 live pairs — against the oracle."""
    from faster_amd import frontend

    vmap = capi.Map(0)
    try:
        whole, faces, info = frontend.forest_batch(65536, seed=5, n_seg=15, max_poly=8, front="device", ctx=ctx, vmap=vmap)
    finally:
        vmap.close()
    assert len(whole) > 60000
    idx = np.sort(np.random.default_rng(7).choice(len(whole), 4096, replace=False))
    wref, oks = check_pairs_against_oracle(ctx, oracle, whole, faces, 15, idx, n_safe=2048)
    assert wref["solved"].mean() > 0.8


def test_independent_model_on_64_gpu_results_at_n10(ctx):
    """SciPy SLSQP on the reference's own unreduced 12N-coefficient model (oracle/py_model.py: rows written one by one as
    solverGurobi.cpp adds them — no jerk space, no reduced space, no active-set code shared with the kernel) under the GPU's
    assignment, for 64 solved C4-sized problems (N = 10, <= 6 polytopes): same cost (1e-6) and the same polynomial (1e-5)."""
    from oracle import py_model

    pr, faces, _ = corridor.whole_batch(96, seed=401, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    res = ctx.solve_batch(pr, faces)
    done = 0
    for i in np.nonzero(res["solved"])[0][:64]:
        p, r = pr[i], res[i]
        fb = int(p["face_begin"])
        polys = [(faces["a"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy(), faces["b"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy())
                 for q in range(int(p["n_poly"]))]
        s = py_model.solve_fixed(10, float(r["dt"]), p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), True, polys,
                                 [int(a) for a in r["assign"][:10]])
        assert s is not None, i
        assert s[0] == pytest.approx(r["cost"], rel=1e-6, abs=1e-7), (i, s[0], r["cost"])
        np.testing.assert_allclose(s[1], r["coeff"][:10], atol=1e-5)
        done += 1
    assert done == 64


def test_independent_model_on_gpu_results_at_n15_and_on_safe_problems(ctx):
    """The same independent check (SciPy SLSQP on the unreduced 12N-coefficient model, oracle/py_model.py) where the first one does not
    reach: 12 solved N = 15 problems with up to 8 polytopes (the C5 kernel instantiation) and 32 solved SAFE problems (no final position
    row, solverGurobi.cpp:343-356 with forceFinalConstraint_ false; x0 = the R the device's hand-off chose) of fused C4 pairs."""
    from oracle import py_model

    def polys_of(p, faces):
        fb = int(p["face_begin"])
        return [(faces["a"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy(), faces["b"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy())
                for q in range(int(p["n_poly"]))]

    def check(p, faces, r, n_seg, force):
        s = py_model.solve_fixed(n_seg, float(r["dt"]), p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), force, polys_of(p, faces),
                                 [int(a) for a in r["assign"][:n_seg]])
        assert s is not None
        assert s[0] == pytest.approx(r["cost"], rel=1e-6, abs=1e-7), (s[0], r["cost"])
        np.testing.assert_allclose(s[1], r["coeff"][:n_seg], atol=1e-5)

    pr, faces, _ = corridor.whole_batch(32, seed=402, n_seg=15, p_choices=(4, 5, 6, 7, 8))
    res = ctx.solve_batch(pr, faces)
    big = np.nonzero(res["solved"])[0][:12]
    assert len(big) == 12
    for i in big:
        check(pr[i], faces, res[i], 15, True)
    whole, wfaces, _ = corridor.whole_batch(96, seed=403, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    wres, sres, safe, sfaces = fused_pairs(ctx, whole, wfaces, corridor.safe_templates(whole), 10, 0.05)
    ok = np.nonzero((safe["n_seg"] > 0) & (sres["solved"] == 1))[0][:32]
    assert len(ok) == 32
    for j in ok:
        assert safe[j]["force_final_pos"] == 0
        check(safe[j], sfaces, sres[j], 10, False)


# ---- row N1 on the device against the reference's OWN sources (oracle/_ref/libref_frontend.so: untouched DecompUtil / jps3d behind
# test-only shims, oracle/ref_frontend/; built in the development container, travels with the snapshot) ----
@pytest.fixture(scope="module")
def ref():
    from oracle.ref_frontend import ref as r

    if r.build() is None:
        pytest.skip("oracle/_ref/libref_frontend.so is not available")
    return r


def test_device_decomposition_against_the_reference_sources(ctx, ref):
    """fh_decompose_batch (K4) against the reference's EllipsoidDecomp3D / LineSegment / DecompBase themselves, driven as
    JPS_Manager::cvxEllipsoidDecomp drives them: same polytopes as sets of rows (1e-9) on the reference's own test path and on
    random scenes — 40+ polytopes."""
    key = lambda M: M[np.lexsort(np.round(M, 6).T[::-1])]
    total = 0
    scenes = [(np.array([[5, 11.5, 0.5], [13, 11.5, 3.0], [14, 10.5, 1.5], [14, 5, 2.5]]), 1, 4000, 0.5)]   # decomp_test_node/data/path3d.txt
    rng0 = np.random.default_rng(17)
    for k in range(10):
        path = np.cumsum(np.vstack([rng0.uniform(-3, 3, 3) * [1, 1, 0] + [0, 0, 1.2], rng0.uniform(0.8, 2.5, (4, 1)) * (rng0.normal(size=(4, 3)) * [1, 1, 0.2])]), axis=0)
        path[:, 2] = np.clip(path[:, 2], 0.6, 2.4)
        scenes.append((path, 100 + k, 1500, 0.4))
    for path, seed, n_cloud, clearance in scenes:
        rng = np.random.default_rng(seed)
        cloud = rng.uniform(path.min(0) - 2.5, path.max(0) + 2.5, size=(n_cloud, 3))
        keep = np.ones(len(cloud), bool)
        for a, b in zip(path[:-1], path[1:]):
            t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
            keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > clearance
        cloud = cloud[keep]
        segs = np.hstack([path[:-1], path[1:]])
        faces, counts = ctx.decompose_batch(cloud, segs, drone_radius=0.05, z_ground=0.0, max_faces=96)
        want = ref.decompose(path, cloud, 0.05, 0.0)
        for i, (A, b) in enumerate(want):
            assert counts[i] == len(b), (seed, i, counts[i], len(b))
            got = np.column_stack([faces["a"][i, :counts[i]], faces["b"][i, :counts[i]]])
            np.testing.assert_allclose(key(got), key(np.column_stack([A, b])), rtol=0, atol=1e-9)
            total += 1
    assert total >= 40


def test_device_map_and_path_search_against_the_reference_sources(ctx, ref):
    """fh_map_read against MapUtil::readMap itself (dimensions, origin, every cell), and fh_map_plan_batch against jps3d driven as
    JPS_Manager::solveJPS3D drives it: a path exists for the same queries, it starts and ends on the same points; the share of
    identical vertex lists is reported (jump point search and this A* return different equal-cost paths for most queries)."""
    from faster_amd import frontend

    cloud, cells, center, starts, goals = frontend.forest_queries(192, 3)
    cloud = cloud.astype(np.float32).astype(np.float64)   # pcl::PointXYZ holds floats
    res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
    rm = ref.Map(cloud, cells, res, center, zg, zmax, infl)
    dm = capi.Map(0)
    try:
        dm.read(cloud, cells, res, center, zg, zmax, infl)
        occ = dm.occupancy()
        assert occ.shape == rm.occupancy().shape and np.array_equal(occ > 0, rm.occupancy() > 0)
        dp, dn, _ = dm.plan_batch(starts, goals)
    finally:
        dm.close()
    same = with_path = 0
    for i in range(len(starts)):
        p, cost, _ = rm.plan(starts[i], goals[i], True)
        assert (p is None) == (dn[i] == 0), i
        if p is None:
            continue
        with_path += 1
        d = dp[i, :dn[i]]
        np.testing.assert_allclose(d[0], p[0], atol=1e-12)
        np.testing.assert_allclose(d[-1], p[-1], atol=1e-12)
        same += int(len(d) == len(p) and np.allclose(d, p, atol=1e-9))
    rm.close()
    assert with_path >= 180 and same >= with_path // 10


@pytest.mark.parametrize("n_seg,p_choices,r_known", [(10, (2, 3, 4, 5, 6), 3.0), (15, (4, 5, 6, 7, 8), 4.0)])
def test_reference_rule_for_r_on_the_device(ctx, oracle, n_seg, p_choices, r_known):
    """fh_set_pair_rule mode 1: R chosen per pair as Faster::findIndexH / findIndexR choose it (faster.cpp:173-251; unknown space
    modelled as everything farther than r_known from the start) inside the fused pair kernel, against the literal numpy restatement
    of the two functions on the oracle's samples (oracle/pair_glue.py): the same pairs skip the safe trajectory, the same sample
    becomes R (state equal to 1e-9), and the safe results match the oracle's on the device-written problems."""
    from oracle import pair_glue

    B = 768
    whole, faces, _ = corridor.whole_batch(B, seed=500 + n_seg, n_seg=n_seg, p_choices=p_choices)
    tmpl = corridor.safe_templates(whole)
    rule = dict(r_known=r_known, drone_radius=0.3, delta_h=1.0, delta_a=0.5)
    ctx.set_pair_rule(mode=1, **rule)
    try:
        wres, sres, safe, sfaces = fused_pairs(ctx, whole, faces, tmpl, n_seg, 0.05)
    finally:
        ctx.set_pair_rule(mode=0)
    wref = oracle.solve_batch(whole, faces)
    compare(wres, wref)
    safe_ref, _ = pair_glue.glue(whole, wref, faces, tmpl, 0.5, 0.2, 3, r_margin=0.05, rule=rule)
    assert np.array_equal(safe["n_seg"], safe_ref["n_seg"])
    live = np.nonzero(safe_ref["n_seg"] > 0)[0]
    assert 0.3 * B < len(live) <= B and (safe_ref["n_seg"] == 0).sum() >= 0
    np.testing.assert_allclose(safe["x0"][live], safe_ref["x0"][live], rtol=0, atol=1e-9)
    assert np.array_equal(safe["n_poly"][live], safe_ref["n_poly"][live])
    sref = oracle.solve_batch(safe[live], sfaces)
    oks = compare(sres[live], sref)
    # R was chosen so that the vehicle can still brake: far more safe problems have a solution than with R at half of the trajectory
    assert oks.mean() > 0.85, oks.mean()


def test_packed_results_are_lossless(ctx):
    """fh_pack_results_device: the records that cross PCIe / xGMI (no dead coefficient rows: 64 + 96 N bytes instead of 1600) unpack
    to exactly the fh_result records of the launch, for every kernel size; the host-side packer agrees byte for byte."""
    import torch

    for n_seg in (6, 10, 15, 16):
        pr, faces, _ = corridor.whole_batch(257, seed=600 + n_seg, n_seg=n_seg, p_choices=(2, 3))
        d_pr, d_fc = _dev(pr), _dev(faces)
        d_res = torch.zeros(len(pr) * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
        mf = int(pr["face_off"][np.arange(len(pr)), pr["n_poly"]].max())
        ctx.solve_batch_device(d_pr.data_ptr(), d_fc.data_ptr(), len(pr), n_seg, mf, d_res.data_ptr())
        pk = capi.packed_result_size(n_seg)
        assert pk == 64 + 96 * n_seg
        d_pack = torch.zeros(len(pr) * pk, dtype=torch.uint8, device="cuda:0")
        ctx.pack_results_device(d_res.data_ptr(), len(pr), n_seg, d_pack.data_ptr())
        ctx.sync()
        full = d_res.cpu().numpy().view(abi.result_dtype)
        packed = d_pack.cpu().numpy()
        back = capi.unpack_results(packed, len(pr), n_seg)
        assert full["solved"].sum() > 100
        assert np.array_equal(back.view(np.uint8), np.ascontiguousarray(full).view(np.uint8))
        assert np.array_equal(capi.pack_results(full, n_seg), packed)


def _plans_equal(host, dev, what):
    hp, hn, hex_ = host
    dp, dn, dex = dev
    assert np.array_equal(hn, dn), (what, np.nonzero(hn != dn)[0][:8])
    assert np.array_equal(hex_, dex), (what, "jump point search on the device popped other nodes than the host")
    for i in np.nonzero(hn > 0)[0]:
        assert np.array_equal(dp[i, :hn[i]], hp[i, :hn[i]]), (what, i)


@pytest.fixture()
def host_jps():
    from faster_amd import frontend

    frontend.set_search("jps")
    yield frontend
    frontend.set_search("astar")


def test_device_jump_point_search_equals_host_restatement(host_jps):
    """fh_map_set_search(1): jump point search in jps3d's own order on the device == plan_path_jps (which is pinned to the reference's
    compiled jps3d vertex for vertex, tests/test_ref_frontend.py): the same number of popped nodes and the same vertices BIT FOR BIT,
    on the forest of config C5 (4096 queries), with the vertices Faster::replan would decompose, after switching the map object
    between the two searches, and on the edge cases of the A* test (wall, empty map, coarse cells, z clipping, 0.1 m cells)."""
    frontend = host_jps
    res, infl, zmax = 0.2, 0.3, 3.0
    cloud, cells, center, starts, goals = frontend.forest_queries(4096, 23)
    m = capi.Map(0)
    try:
        m.read(cloud, cells, res, center, 0.0, zmax, infl)
        astar = m.plan_batch(starts[:512], goals[:512])
        m.set_search("jps")
        host = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals)
        assert (host[1] > 0).mean() > 0.95 and host[2].mean() > 100
        _plans_equal(host, m.plan_batch(starts, goals), "forest")
        _plans_equal(host, m.plan_batch(starts, goals), "forest, second call (serial numbers)")
        _plans_equal(frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8),
                     m.plan_batch(starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8), "refined")
        m.set_search("astar")
        again = m.plan_batch(starts[:512], goals[:512])
        assert np.array_equal(astar[1], again[1]) and np.array_equal(astar[2], again[2])
        m.set_search("jps")
        rng = np.random.default_rng(5)
        for case in ["wall", "empty", "coarse", "tall", "large"]:
            res, infl, zg, zmax = 0.25, 0.25, 0.0, 2.0
            cells, center = (40, 40, 8), np.array([5.0, 5.0, 1.0])
            if case == "wall":
                yy, zz = np.meshgrid(np.arange(-3, 13, 0.1), np.arange(-1, 3, 0.1))
                cloud = np.column_stack([np.full(yy.size, 5.0), yy.ravel(), zz.ravel()])
            elif case == "empty":
                cloud = np.zeros((0, 3))
            elif case == "coarse":
                cloud, _ = frontend.forest_cloud(4, size=(10.0, 10.0, 2.0), density=0.2)
                res, infl, cells = 0.5, 0.2, (20, 20, 4)
            elif case == "tall":
                cloud, _ = frontend.forest_cloud(5, size=(10.0, 10.0, 2.0), density=0.2)
                center, cells = np.array([5.0, 5.0, 1.7]), (40, 40, 12)
            else:
                cloud, _ = frontend.forest_cloud(8, size=(10.0, 10.0, 2.0), density=0.25)
                res, infl, cells = 0.1, 0.3, (100, 100, 30)
            n = 256
            starts = np.column_stack([rng.uniform(0.5, 4.0, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
            goals = np.column_stack([rng.uniform(6.0, 9.5, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
            goals[:8] = starts[:8] + 0.01
            goals[8] = starts[8]
            starts[9] = [-50.0, 5.0, 1.0]
            goals[10] = [5.0, 500.0, 1.0]
            starts[11, 2] = -0.7
            goals[12, 2] = -0.2
            starts[13] = goals[13] - [0.3, 0.0, 0.0]
            m.read(cloud, cells, res, center, zg, zmax, infl)
            host = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals)
            dev = m.plan_batch(starts, goals)
            _plans_equal(host, dev, case)
            assert dev[1][9] == 0 and dev[1][10] == 0
            if case == "wall":
                assert (dev[1][16:] == 0).all()
            if case == "empty":
                assert (dev[1][16:] == 2).all()
    finally:
        m.close()


def test_device_jump_point_search_against_the_reference_sources(ref):
    """The device search in mode 1 against jps3d ITSELF (the reference's sources compiled untouched, driven as JPS_Manager::solveJPS3D
    drives them): the same queries have a path and every vertex list is identical."""
    from faster_amd import frontend

    total = 0
    for seed in (3, 4):
        cloud, cells, center, starts, goals = frontend.forest_queries(160, seed)
        cloud = cloud.astype(np.float32).astype(np.float64)   # pcl::PointXYZ holds floats
        res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
        rm = ref.Map(cloud, cells, res, center, zg, zmax, infl)
        dm = capi.Map(0)
        try:
            dm.read(cloud, cells, res, center, zg, zmax, infl)
            dm.set_search("jps")
            dp, dn, _ = dm.plan_batch(starts, goals, max_points=128)
        finally:
            dm.close()
        for i in range(len(starts)):
            p, _, _ = rm.plan(starts[i], goals[i], True)
            assert (p is None) == (dn[i] == 0), (seed, i)
            if p is None:
                continue
            assert dn[i] == len(p), (seed, i, dn[i], len(p))
            np.testing.assert_allclose(dp[i, :dn[i]], p, rtol=0, atol=1e-9)
            total += 1
        rm.close()
    assert total >= 300


def test_device_corridor_front_end_with_jump_point_search_equals_host(ctx):
    """Config C5's whole front-end (map -> jump point search in jps3d's order -> createMoreVertexes / deleteVertexes -> decomposition)
    on the device against the CPU front-end run with plan_path_jps: the same pairs kept, every problem field and every polytope row
    bit for bit.  These are the corridors FASTER itself would build (its own path, its own decomposition)."""
    from faster_amd import frontend

    n = 3072
    hp, hf, hi = frontend.forest_batch(n, 33, search="jps")
    vmap = capi.Map(0)
    try:
        dp, df, di = frontend.forest_batch(n, 33, front="device", ctx=ctx, vmap=vmap, search="jps")
    finally:
        vmap.close()
    assert np.array_equal(hi["kept"], di["kept"]) and len(hp) > 0.95 * n
    for f in abi.problem_dtype.names:
        assert np.array_equal(hp[f], dp[f]), f
    assert hf.shape == df.shape and np.array_equal(hf["a"], df["a"]) and np.array_equal(hf["b"], df["b"])
    ap, _, _ = frontend.forest_batch(256, 33)   # (the A* corridors differ: another of the equal-cost paths)
    assert not np.array_equal(ap["face_off"], hp["face_off"][:len(ap)]) or len(ap) != 256


@pytest.mark.parametrize("kind", ["near_trees", "blobs_3d", "tight_cubes"])
def test_device_jump_point_search_where_the_freed_cubes_and_3d_jumps_matter(host_jps, kind):
    """The cases the jump tables have to hand over to the cell-by-cell evaluation: starts and goals INSIDE the inflated hull of a tree
    (the cubes freed around them hold occupied cells, so entries near them do not hold), a map as high as it is wide with random
    blobs (space-diagonal jumps, goals above and below), and a coarse inflation (big cubes) — device == plan_path_jps bit for bit."""
    frontend = host_jps
    rng = np.random.default_rng(11)
    if kind == "near_trees":
        cloud, centres = frontend.forest_cloud(3)
        res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
        cells, center = (110, 110, 15), np.array([10.0, 10.0, 1.5])
        n = 1536
        ang = rng.uniform(0, 2 * np.pi, n)
        near = centres[rng.integers(0, len(centres), n)] + np.column_stack([np.cos(ang), np.sin(ang)]) * rng.uniform(0.40, 0.75, (n, 1))
        starts = np.column_stack([near, rng.uniform(0.8, 2.2, n)])
        goals = np.column_stack([rng.uniform(2, 18, n), rng.uniform(2, 18, n), rng.uniform(0.8, 2.2, n)])
        starts[n // 2:], goals[n // 2:] = goals[n // 2:].copy(), starts[n // 2:].copy()
    else:
        res, zg, zmax = 0.25, 0.0, 10.0
        infl = 0.25 if kind == "blobs_3d" else 0.6
        cells, center = (40, 40, 40), np.array([5.0, 5.0, 5.0])
        blobs = rng.uniform(0.5, 9.5, (60, 3))
        cloud = (blobs[:, None, :] + rng.normal(0, 0.25, (60, 40, 3))).reshape(-1, 3)
        n = 1536
        starts = rng.uniform(0.3, 9.7, (n, 3))
        goals = rng.uniform(0.3, 9.7, (n, 3))
    host = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, max_points=128)
    m = capi.Map(0)
    try:
        m.read(cloud, cells, res, center, zg, zmax, infl)
        m.set_search("jps")
        dev = m.plan_batch(starts, goals, max_points=128)
    finally:
        m.close()
    assert (host[1] > 0).mean() > 0.8
    _plans_equal(host, dev, kind)


@pytest.mark.parametrize("mode", [0, 1])
def test_plans_appended_on_the_device(ctx, oracle, mode):
    """fh_append_plans_device = Faster::appendToPlan (faster.cpp:606-648) for a batch of pairs: the committed plan is the whole
    trajectory's samples 0 .. k_safe followed by the safe trajectory's samples — against the numpy restatement on the ORACLE's
    samples (oracle/pair_glue.py append_plans): same pairs commit, same k_safe, same lengths, states equal to 1e-9; and bit for bit
    against the device's own fillX (fh_sample_batch_device) spliced on the host.  mode 1: FASTER's rule for R (pairs whose whole
    trajectory stays in known space commit the whole trajectory)."""
    import torch

    from oracle import pair_glue

    B, n_seg = 512, 10
    whole, faces, _ = corridor.whole_batch(B, seed=910 + mode, n_seg=n_seg, p_choices=(2, 3, 4, 5))
    tmpl = corridor.safe_templates(whole)
    goal_dist = np.linalg.norm(whole["xf"][:, :3] - whole["x0"][:, :3], axis=1)
    rule = dict(r_known=float(np.median(goal_dist)) + 0.3, drone_radius=0.3, delta_h=1.0, delta_a=0.5)   # half of the goals in known space
    ctx.set_pair_rule(mode=mode, **rule)
    try:
        wres, sres, safe, sfaces = fused_pairs(ctx, whole, faces, tmpl, n_seg, 0.05)
        d_w, d_wr, d_s, d_sr = _dev(whole), _dev(wres), _dev(safe), _dev(sres)
        max_states = 1024
        d_plans = torch.zeros(B * max_states * abi.state_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
        d_counts = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        d_k = torch.zeros(B, dtype=torch.int32, device="cuda:0")
        ctx.append_plans_device(d_w.data_ptr(), d_wr.data_ptr(), d_s.data_ptr(), d_sr.data_ptr(), B, 0.5, max_states, d_plans.data_ptr(),
                                d_counts.data_ptr(), d_k.data_ptr())
        # the device's own samples of both trajectories
        d_xw = torch.zeros_like(d_plans); d_cw = torch.zeros_like(d_counts)
        d_xs = torch.zeros_like(d_plans); d_cs = torch.zeros_like(d_counts)
        ctx.sample_batch_device(d_w.data_ptr(), d_wr.data_ptr(), B, max_states, d_xw.data_ptr(), d_cw.data_ptr())
        ctx.sample_batch_device(d_s.data_ptr(), d_sr.data_ptr(), B, max_states, d_xs.data_ptr(), d_cs.data_ptr())
        ctx.sync()
    finally:
        ctx.set_pair_rule(mode=0)
    plans = d_plans.cpu().numpy().view(abi.state_dtype).reshape(B, max_states)
    counts, ks = d_counts.cpu().numpy(), d_k.cpu().numpy()
    xw = d_xw.cpu().numpy().view(abi.state_dtype).reshape(B, max_states)
    xs = d_xs.cpu().numpy().view(abi.state_dtype).reshape(B, max_states)
    cw, cs = d_cw.cpu().numpy(), d_cs.cpu().numpy()
    ref_plans, ref_k = pair_glue.append_plans(whole, wres, safe, sres, 0.5, rule if mode == 1 else None)
    assert np.array_equal(ks, ref_k)
    committed = 0
    for i in range(B):
        if ref_plans[i] is None:
            assert counts[i] == 0, i
            continue
        committed += 1
        assert counts[i] == len(ref_plans[i]) <= max_states, (i, counts[i], len(ref_plans[i]))
        got = plans[i, :counts[i]]
        for f in ("pos", "vel", "accel", "jerk"):
            np.testing.assert_allclose(got[f], ref_plans[i][f], rtol=0, atol=1e-9)
        k = ks[i]
        n_safe = counts[i] - (k + 1)
        assert n_safe == (cs[i] if safe["n_seg"][i] > 0 else 0)
        assert got[:k + 1].tobytes() == xw[i, :k + 1].tobytes() and got[k + 1:].tobytes() == xs[i, :n_safe].tobytes()
        assert k <= cw[i] - 1
    assert committed > 0.4 * B
    if mode == 1:
        whole_only = (safe["n_seg"] == 0) & (wres["solved"] == 1)
        assert whole_only.any() and np.array_equal(counts[whole_only], cw[whole_only])   # no unknown space on the way: the whole trajectory


@pytest.mark.parametrize("r_known", [4.0, 100.0, 0.2])
def test_safe_corridor_decomposed_around_r_on_the_device(ctx, oracle, r_known, seed=41, n_check=40):
    """fh_safe_corridor_batch_device: the safe corridor of Faster::replan (faster.cpp:446-524) — JPS_in cut at unknown space, R first,
    decomposition against unknown + occupied points, xf = G or M — for forest pairs whose corridors and whole trajectories come from
    the device.  Against oracle/pair_glue.py: the safe paths equal the restatement (1e-9); the polytopes equal the HOST decomposition
    of the same paths against the explicit cloud [unknown voxels of the map's grid, z-major | occupied points] BIT FOR BIT; xf follows
    the G-inside rule; the safe problems then solve on the device like the oracle's."""
    import torch

    from faster_amd import frontend
    from oracle import pair_glue

    # whole corridors as FASTER builds them: at most 3 polytopes of <= 1.5 m (max_poly_whole 3), so that the whole trajectory comes to
    # rest about as far out as the vehicle has seen (Ra = r_known = 4 m) — with the 12 m corridors of config C5 the vehicle is still
    # fast where known space ends and no safe trajectory exists inside it (v_max 5, a_max 5, j_max 8: all 96 infeasible, as they should be)
    # r_known 100: nobody comes near unknown space (no safe problem at all); 0.2 < drone_radius: the start itself is at the boundary —
    # the reference's 1 cm stub path, a corridor squeezed between unknown voxels
    n, N, max_poly, mps = (128 if r_known == 4.0 else 48), 10, 3, 3
    res, infl, zmax, drone_r, decomp_r = 0.2, 0.3, 3.0, 0.3, 0.05
    vmap = capi.Map(0)
    try:
        pr, fc, info = frontend.forest_batch(n, seed, n_seg=N, max_poly=max_poly, front="device", ctx=ctx, vmap=vmap, search="jps", sphere_ra=4.0)
        cloud, cells, center, starts, goals = frontend.forest_queries(n, seed)
        vmap.set_sphere(4.0)   # JPS_in: the path inside the sphere Ra (faster.cpp:370-382), as for the whole corridor above
        paths, npts, _ = vmap.plan_batch(starts, goals, max_points=max_poly + 1, max_vertex_dist=1.5, max_poly=max_poly)
        dims, origin = vmap.dims()
    finally:
        vmap.close()
    kept = info["kept"]
    paths, npts, goals = paths[kept], npts[kept], goals[kept]
    B = len(pr)
    assert B > 0.9 * n and (npts >= 2).all()
    mf = int(pr["face_off"][np.arange(B), pr["n_poly"]].max())
    d_pr, d_fc = _dev(pr), _dev(fc)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    ctx.solve_batch_device(d_pr.data_ptr(), d_fc.data_ptr(), B, N, mf, d_wr.data_ptr())
    tmpl = corridor.safe_templates(pr)
    tmpl["n_seg"] = N
    fpp = 96
    d_safe = _dev(tmpl)
    d_sf = torch.zeros(B * fpp * abi.face_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_paths, d_np, d_goals, d_cloud = _dev(paths), _dev(npts.astype(np.int32)), _dev(goals), _dev(cloud)
    d_sp = torch.zeros(B * (mps + 1) * 3, dtype=torch.float64, device="cuda:0")
    d_snp = torch.zeros(B, dtype=torch.int32, device="cuda:0")
    rule = dict(r_known=r_known, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
    ctx.set_pair_rule(mode=1, **rule)
    try:
        ctx.safe_corridor_batch_device(d_pr.data_ptr(), d_wr.data_ptr(), d_paths.data_ptr(), d_np.data_ptr(), max_poly + 1, d_goals.data_ptr(),
                                       d_cloud.data_ptr(), len(cloud), origin, res, dims, B, 0.5, mps, (2.0, 2.0, 1.0), decomp_r, 0.0, fpp, N,
                                       d_safe.data_ptr(), d_sf.data_ptr(), d_sp.data_ptr(), d_snp.data_ptr())
        ctx.sync()
    finally:
        ctx.set_pair_rule(mode=0)
    wres = d_wr.cpu().numpy().view(abi.result_dtype)
    safe = d_safe.cpu().numpy().view(abi.problem_dtype)
    sfaces = d_sf.cpu().numpy().view(abi.face_dtype).reshape(B, fpp)
    spaths, snp = d_sp.cpu().numpy().reshape(B, mps + 1, 3), d_snp.cpu().numpy()
    assert (wres["solved"] == 1).mean() > 0.9
    live = np.nonzero(snp >= 2)[0]
    if r_known == 100.0:
        assert len(live) == 0 and (safe["n_seg"] == 0).all()
        return
    if r_known == 0.2:
        assert len(live) == (wres["solved"] == 1).sum()
        assert np.allclose(spaths[live, 1], pr["x0"][live, :3] + [0.01, 0.0, 0.0], atol=1e-12) and (snp[live] == 2).all()
    else:
        assert 0.4 * B < len(live) < B        # (the others never come near unknown space: no safe trajectory needed)
    assert np.array_equal(safe["n_seg"][snp < 2], np.zeros((snp < 2).sum(), dtype=safe["n_seg"].dtype))
    checked = 0
    for i in live[:n_check]:
        A = pr["x0"][i, :3]
        want = pair_glue.safe_path(paths[i, :npts[i]], A, safe["x0"][i, :3], r_known, drone_r, mps)
        assert len(want) == snp[i], (i, len(want), snp[i])
        np.testing.assert_allclose(spaths[i, :snp[i]], want, rtol=0, atol=1e-9)
        # known space only: every vertex of the safe path but R's successor chain ends before unknown space
        assert r_known < 1.0 or np.linalg.norm(spaths[i, snp[i] - 1] - A) <= r_known + 1e-6
        full = np.vstack([pair_glue.unknown_voxels(origin, res, dims, A, r_known), cloud])
        polys, _ = frontend.decompose(spaths[i, :snp[i]], full, drone_radius=decomp_r, z_ground=0.0, bbox=(2.0, 2.0, 1.0), max_faces=4096)
        P = int(safe["n_poly"][i])
        assert safe["n_seg"][i] == N and P == len(polys) == snp[i] - 1 and safe["face_begin"][i] == i * fpp
        for p, (Ah, bh) in enumerate(polys):
            f0, f1 = safe["face_off"][i, p], safe["face_off"][i, p + 1]
            assert f1 - f0 == len(bh), (i, p, f1 - f0, len(bh))
            assert np.array_equal(sfaces["a"][i, f0:f1], Ah) and np.array_equal(sfaces["b"][i, f0:f1], bh), (i, p)
        Al, bl = polys[-1]
        inside = not np.any(Al @ goals[i] - bl > 0)
        assert np.array_equal(safe["xf"][i, :3], goals[i] if inside else spaths[i, snp[i] - 1]), i
        # R inside its first polytope (the corridor is decomposed around it)
        A0, b0 = polys[0]
        assert r_known < 1.0 or np.all(A0 @ safe["x0"][i, :3] - b0 <= 1e-9)
        checked += 1
    assert checked >= min(30, n_check)
    if r_known < 1.0:
        return
    # the safe problems solve, and as the oracle solves them
    sel = live
    sp = safe[sel].copy()
    d_sp2 = _dev(sp)
    d_sr = torch.zeros(len(sel) * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    ctx.solve_batch_device(d_sp2.data_ptr(), d_sf.data_ptr(), len(sel), N, fpp, d_sr.data_ptr())
    ctx.sync()
    sres = d_sr.cpu().numpy().view(abi.result_dtype)
    compare(sres[:32], oracle.solve_batch(sp[:32], sfaces.reshape(-1)))
    assert (sres["solved"] == 1).mean() > 0.8, ((sres["solved"] == 1).mean(), np.unique(sres["status"], return_counts=True), sp[0], sres[0]["trials"])


def test_device_paths_clipped_to_the_sphere_equal_host(host_jps):
    """fh_map_set_sphere: JPS_in of Faster::replan (the path cut at the sphere of radius min(|goal - start| - 0.001, Ra) around the
    start, crossing point appended, then createMoreVertexes / deleteVertexes) on the device == the host front-end bit for bit."""
    frontend = host_jps
    cloud, cells, center, starts, goals = frontend.forest_queries(2048, 10)
    res, zmax, infl, Ra = 0.2, 3.0, 0.3, 4.0
    m = capi.Map(0)
    try:
        m.read(cloud, cells, res, center, 0.0, zmax, infl)
        m.set_search("jps")
        m.set_sphere(Ra)
        frontend.set_sphere(Ra)
        for kw in (dict(max_points=64), dict(max_points=4, max_vertex_dist=1.5, max_poly=3)):
            host = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, **kw)
            dev = m.plan_batch(starts, goals, **kw)
            _plans_equal(host, dev, str(kw))
            ends = np.array([host[0][i, host[1][i] - 1] for i in np.nonzero(host[1] > 0)[0]])
            first = np.array([host[0][i, 0] for i in np.nonzero(host[1] > 0)[0]])
            assert (np.linalg.norm(ends - first, axis=1) <= Ra + 1.5 + 1e-6).all()
        m.set_sphere(0.0)
        frontend.set_sphere(0.0)
        _plans_equal(frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals), m.plan_batch(starts, goals), "off again")
    finally:
        frontend.set_sphere(0.0)
        m.close()


def test_problem_records_from_corridors_on_the_device(ctx):
    """fh_corridor_problems_device (faster.cpp:393-404): polytope table of every pair's record and xf = G when G lies in the last polytope
    of the whole corridor, else the last vertex of the path; pairs without a path get n_seg = 0; the caller's fields are left alone."""
    import torch

    from faster_amd import frontend

    n, N, max_poly, fpp = 512, 6, 3, 96
    cloud, cells, center, starts, goals = frontend.forest_queries(n, 12)
    goals[:64] = starts[:64] + np.array([0.6, 0.4, 0.0])      # goals close by: inside the last (only) polytope
    starts[64] = [-40.0, 3.0, 1.0]                             # outside the map: no path
    vmap = capi.Map(0)
    try:
        vmap.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3)
        vmap.set_search("jps")
        vmap.set_sphere(4.0)
        paths, npts, _ = vmap.plan_batch(starts, goals, max_points=max_poly + 1, max_vertex_dist=1.5, max_poly=max_poly)
    finally:
        vmap.close()
    tmpl = abi.make_problems(n)
    tmpl["n_seg"], tmpl["dc"], tmpl["v_max"], tmpl["f_init"] = 99, 0.01, 5.0, 2.0
    tmpl["x0"][:, :3] = starts
    d_paths, d_np, d_goals, d_cloud, d_pr = _dev(paths), _dev(npts.astype(np.int32)), _dev(goals), _dev(cloud), _dev(tmpl)
    d_f = torch.zeros(n * fpp * abi.face_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_off = torch.zeros((n, 9), dtype=torch.int32, device="cuda:0")
    d_npoly = torch.zeros(n, dtype=torch.int32, device="cuda:0")
    d_last = torch.zeros((n, 3), dtype=torch.float64, device="cuda:0")
    ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), d_paths.data_ptr(), d_np.data_ptr(), n, max_poly + 1, max_poly, fpp, d_f.data_ptr(),
                              d_off.data_ptr(), d_npoly.data_ptr(), d_last.data_ptr())
    ctx.corridor_problems_device(d_np.data_ptr(), d_last.data_ptr(), d_goals.data_ptr(), d_f.data_ptr(), d_off.data_ptr(), d_npoly.data_ptr(), n, fpp, N,
                                 d_pr.data_ptr())
    ctx.sync()
    pr = d_pr.cpu().numpy().view(abi.problem_dtype)
    faces = d_f.cpu().numpy().view(abi.face_dtype).reshape(n, fpp)
    off, npoly, last = d_off.cpu().numpy(), d_npoly.cpu().numpy(), d_last.cpu().numpy()
    assert pr["n_seg"][64] == 0 and npts[64] == 0
    inside_count = 0
    for i in range(n):
        if npts[i] < 2 or npoly[i] < 1:
            assert pr["n_seg"][i] == 0
            continue
        assert pr["n_seg"][i] == N and pr["n_poly"][i] == npoly[i] and pr["face_begin"][i] == i * fpp
        assert np.array_equal(pr["face_off"][i], off[i])
        P = npoly[i]
        A, b = faces["a"][i, off[i, P - 1]:off[i, P]], faces["b"][i, off[i, P - 1]:off[i, P]]
        inside = not np.any(A[:, 0] * goals[i, 0] + A[:, 1] * goals[i, 1] + A[:, 2] * goals[i, 2] - b > 0)
        inside_count += inside
        assert np.array_equal(pr["xf"][i, :3], goals[i] if inside else last[i]), i
        assert np.array_equal(pr["x0"][i], tmpl["x0"][i]) and pr["dc"][i] == 0.01 and pr["f_init"][i] == 2.0
    assert inside_count >= 32 and inside_count < n


@pytest.mark.gpu
@pytest.mark.parametrize("n_seg,p_choices", [(5, (2, 3)), (10, (3, 4, 5)), (15, (4, 5))])
def test_fast_safe_problems_follow_the_oracle_tree(oracle, n_seg, p_choices):
    """Safe problems that start fast and still accelerating in a corridor pulled in by 0.4 m: many trials end outside the corridor at
    their root, so the search branches on the earliest violated segment (below the root, or at it when the overshoot exceeds 1.2
    braking distances).  Results against the oracle; without work sharing the device explores the oracle's tree minus the subtrees
    the conflict sets let it skip (never more nodes), and with it the results are the same bit for bit."""
    pr, faces, verts = corridor.safe_batch(512, seed=600 + n_seg, n_seg=n_seg, p_choices=p_choices)
    faces = faces.copy()
    faces["b"] -= 0.4
    u = verts[:, 1] - verts[:, 0]
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pr["x0"][:, 3:6], pr["x0"][:, 6:9] = 4.8 * u, 2.0 * u
    ref = oracle.solve_batch(pr, faces)
    assert 0.3 < ref["solved"].mean() < 0.98
    par = abi.default_params()
    shared, solo = capi.Context(0), capi.Context(0)
    try:
        shared.set_params(par)
        par["share"] = 0
        solo.set_params(par)
        got, bounded = shared.solve_batch(pr, faces), solo.solve_batch(pr, faces)
        solo.set_sched(child_bound=0)  # every child visited: the oracle's tree (the default skips children that cannot hold a better leaf)
        alone = solo.solve_batch(pr, faces)
    finally:
        shared.close()
        solo.close()
    compare(alone, ref)
    assert np.all(alone["nodes"] <= ref["nodes"]) and (alone["nodes"] == ref["nodes"]).mean() > 0.5
    assert np.all(bounded["nodes"] <= alone["nodes"]) and bounded["nodes"].sum() < alone["nodes"].sum()
    for f in ("solved", "trials", "status", "factor", "dt", "cost", "coeff", "assign"):
        assert np.array_equal(alone[f], got[f]), f
        assert np.array_equal(bounded[f], got[f]), f
