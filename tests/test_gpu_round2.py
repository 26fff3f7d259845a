"""GPU tests added in round 2 (all through the C ABI): oracle comparison at BASELINE's full C4 size, the device decomposition
against the numpy restatement directly, the closed loop against the oracle-backed run, a slice of the randomized parity sweep
(including the tight, mostly infeasible region), one batch over a pool of devices, cooperative cancellation and the deadline."""
import json
import os
import subprocess
import threading
import time

import numpy as np
import pytest
import torch  # noqa: F401  (before libfasterhip.so is loaded: one HIP runtime per process)

from faster_amd import abi, capi, corridor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESULT_FIELDS = [n for n in abi.result_dtype.names if n not in ("nodes", "qp_iters", "kflops")]


@pytest.fixture(scope="module")
def ctx():
    import torch  # noqa: F401  (torch first: one HIP runtime in the process)

    c = capi.Context(0)
    yield c
    c.close()


def compare(got, ref, cost_rtol=1e-7, coeff_atol=1e-6):
    assert np.array_equal(got["solved"], ref["solved"]), np.nonzero(got["solved"] != ref["solved"])
    assert np.array_equal(got["trials"], ref["trials"])
    assert np.array_equal(got["factor"], ref["factor"]) and np.array_equal(got["dt"], ref["dt"])
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["solved"] == 1
    np.testing.assert_allclose(got["cost"][ok], ref["cost"][ok], rtol=cost_rtol, atol=1e-9)
    np.testing.assert_allclose(got["coeff"][ok], ref["coeff"][ok], rtol=0, atol=coeff_atol)
    return ok


def test_full_size_c4_subsample_against_oracle(ctx, oracle):
    """BASELINE config C4 at its full size (32768 whole+safe pairs, both hand-off variants) in ONE fused launch; a random
    subsample of 4096 pairs — whole problem, the hand-off record the device wrote, and the safe problem built from it — is
    compared with the oracle (the full batch would take the CPU minutes)."""
    import torch

    from oracle import pair_glue

    B, N = 32768, 10
    whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    tmpl = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    dev = "cuda:0"

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    d_whole, d_faces = to_dev(whole), to_dev(faces)
    idx = np.sort(np.random.default_rng(5).choice(B, 4096, replace=False))
    for margin in (0.05, -1.0):
        ctx.set_pair_margin(margin)
        d_safe, d_sf = to_dev(tmpl), torch.zeros_like(d_faces)
        d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros_like(d_wr)
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(),
                               d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync()
        wres = d_wr.cpu().numpy().view(abi.result_dtype)
        sres = d_sr.cpu().numpy().view(abi.result_dtype)
        safe = d_safe.cpu().numpy().view(abi.problem_dtype)
        sfaces = d_sf.cpu().numpy().view(abi.face_dtype)
        wref = oracle.solve_batch(whole[idx], faces)
        compare(wres[idx], wref)
        # the hand-off records of the subsample against the host restatement driven by the ORACLE's whole results
        safe_ref, sfaces_ref = pair_glue.glue(whole[idx], wref, faces, tmpl[idx], 0.5, 0.2, 3, r_margin=margin)
        assert np.array_equal(safe["n_seg"][idx], safe_ref["n_seg"]) and np.array_equal(safe["n_poly"][idx], safe_ref["n_poly"])
        np.testing.assert_allclose(safe["x0"][idx], safe_ref["x0"], rtol=0, atol=1e-9)
        live = safe_ref["n_seg"] > 0
        sub = np.nonzero(live)[0][:2048]
        sref = oracle.solve_batch(safe[idx][sub], sfaces)   # the device-written safe problems, solved by the oracle
        ok = compare(sres[idx][sub], sref)
        assert ok.mean() > 0.6
        if margin >= 0:  # R strictly inside its corridor: the first polytope contains x0
            s = safe[idx][sub]
            for k in range(0, len(s), 97):
                f0 = s["face_begin"][k]
                n0 = s["face_off"][k][1]
                assert np.all(sfaces["a"][f0:f0 + n0] @ s["x0"][k][:3] < sfaces["b"][f0:f0 + n0])
    ctx.set_pair_margin(-1.0)


def test_gpu_decomposition_against_numpy_restatement(ctx):
    """fh_decompose_batch against oracle/decomp_oracle.py DIRECTLY (not through the host front-end): polytopes as sets of rows."""
    from oracle import decomp_oracle

    key = lambda M: M[np.lexsort(np.round(M, 7).T[::-1])]
    rng = np.random.default_rng(17)
    total = 0
    for scene in range(5):
        path = np.cumsum(np.vstack([rng.uniform(-3, 3, 3) * [1, 1, 0] + [0, 0, 1.2], rng.uniform(0.8, 2.5, (4, 1)) * (rng.normal(size=(4, 3)) * [1, 1, 0.2])]), axis=0)
        path[:, 2] = np.clip(path[:, 2], 0.6, 2.4)
        cloud = rng.uniform(path.min(0) - 2.5, path.max(0) + 2.5, size=(700, 3))
        keep = np.ones(len(cloud), bool)
        for a, b in zip(path[:-1], path[1:]):
            t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
            keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > 0.45
        cloud = cloud[keep]
        segs = np.hstack([path[:-1], path[1:]])
        faces, counts = ctx.decompose_batch(cloud, segs, drone_radius=0.05, z_ground=0.0, max_faces=64)
        ref = decomp_oracle.decompose_path(path, cloud, 0.05, 0.0)
        for i, (A, b) in enumerate(ref):
            assert counts[i] == len(b), (scene, i, counts[i], len(b))
            got = np.column_stack([faces["a"][i, :counts[i]], faces["b"][i, :counts[i]]])
            np.testing.assert_allclose(key(got), key(np.column_stack([A, b])), atol=1e-9)
            total += 1
    assert total == 20


def test_closed_loop_log_equals_oracle_backed_run():
    """N2: the closed loop driven by SolverHip on the GPU produces the same log as the run in which the two C-ABI calls are
    answered by the oracle (same seed): every counter equal, every metric equal to 1e-6."""
    from test_replan_stub import build

    exe = build()
    logs = {}
    for mode, arg in (("gpu", "-"), ("oracle", os.path.join(ROOT, "oracle", "liboracle.so"))):
        r = subprocess.run([exe, mode, arg, "1"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        logs[mode] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    g, o = logs["gpu"], logs["oracle"]
    assert g.keys() == o.keys()
    for k in g:
        if isinstance(g[k], (int, list)) and not isinstance(g[k], bool) and all(isinstance(v, int) for v in (g[k] if isinstance(g[k], list) else [g[k]])):
            assert g[k] == o[k], (k, g[k], o[k])
        else:
            np.testing.assert_allclose(g[k], o[k], rtol=0, atol=1e-6, err_msg=k)


def test_parity_sweep_slice(ctx, oracle):
    """~40 s of the randomized sweep of tests/tools/parity_sweep.py (N in {3..15}, P <= 8, whole / safe, bounds, increments),
    INCLUDING tightened corridors with N*P > 40 (mostly infeasible: the exact search has no incumbent to prune with)."""
    rng = np.random.default_rng(2027)
    t0 = time.time()
    done = tight_big = 0
    cfg = 0
    while time.time() - t0 < 40.0 or tight_big < 2:
        cfg += 1
        n_seg = int(rng.choice([3, 5, 6, 8, 10, 12, 15]))
        pmax = int(rng.integers(1, 9))
        if cfg in (2, 5):  # make sure the tight many-segments-times-many-polytopes region is visited
            n_seg, pmax = (10, 6) if cfg == 2 else (12, 5)
        pch = tuple(range(max(1, pmax - 2), pmax + 1))
        kw = dict(speed=float(rng.uniform(0.5, 4.5)), lateral=float(rng.uniform(0.0, 1.5)), acc0=float(rng.uniform(0, 3)),
                  f_inc=float(rng.choice([0.5, 1.0, 1.0, 2.0])), v_max=float(rng.choice([3, 5])), a_max=float(rng.choice([3, 5])),
                  j_max=float(rng.choice([5, 8])))
        tight = rng.random() < 0.3 or cfg in (2, 5)
        big = n_seg * pmax > 40
        n = (24 if big and tight else 768) if n_seg * pmax <= 60 else (192 if n_seg * pmax <= 80 else 48)
        force = bool(rng.random() < 0.6)
        pr, faces, _ = corridor.make_batch(n, n_seg, pch, force, int(rng.integers(1 << 30)), **kw)
        if tight:
            faces = faces.copy()
            faces["b"] -= rng.uniform(0.3, 0.6)
            tight_big += 1 if big else 0
        got = ctx.solve_batch(pr, faces)
        ref = oracle.solve_batch(pr, faces)
        compare(got, ref)
        done += n
        if time.time() - t0 > 150:
            break
    assert done >= 1000 and tight_big >= 2


def test_pool_shards_equal_one_way(ctx):
    """ONE batch over a pool of devices (fh_pool_*): contiguous blocks, host scatter, gather of complete fh_result blocks into
    the host array and, with peer copies, into the root device's memory.  The box has one GPU, so the pool names it three
    times (three contexts, three streams, three shards of 37: 13 + 13 + 11): the partition / rebasing / gather logic is the
    same as with three devices, and the results equal the one-way run record for record."""
    import torch

    pr, faces, _ = corridor.whole_batch(37, seed=4, n_seg=10, p_choices=(2, 3, 4, 5))
    one = ctx.solve_batch(pr, faces)
    pool = capi.Pool([0, 0, 0])
    assert pool.size() == 3
    d_root = torch.zeros(37 * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    got = pool.solve_batch(pr, faces, root=1, d_results_root=d_root.data_ptr())
    torch.cuda.synchronize()
    on_root = d_root.cpu().numpy().view(abi.result_dtype)
    for f in RESULT_FIELDS:
        assert np.array_equal(got[f], one[f]), f
        assert np.array_equal(on_root[f], one[f]), ("root", f)
    # pairs: whole -> hand-off -> safe per block
    tmpl = corridor.safe_templates(pr)
    pool.set_pair_margin(0.05)
    w, s = pool.solve_pairs(pr, faces, tmpl)
    pool1 = capi.Pool([0])
    pool1.set_pair_margin(0.05)
    w1, s1 = pool1.solve_pairs(pr, faces, tmpl)
    for f in RESULT_FIELDS:
        assert np.array_equal(w[f], one[f]) and np.array_equal(w1[f], one[f]), f
        assert np.array_equal(s[f], s1[f]), ("safe", f)
    assert s["solved"].sum() > 10
    # a pool smaller than the batch count and larger than it
    for devs, n in (([0, 0], 1), ([0, 0, 0, 0], 3)):
        p2 = capi.Pool(devs)
        g2 = p2.solve_batch(pr[:n], faces)
        for f in RESULT_FIELDS:
            assert np.array_equal(g2[f], one[f][:n]), f
        p2.close()
    pool.close()
    pool1.close()


# ---- voxel map + batched path search on the device (fh_map_*), SURVEY.md 8(f) N1 first half -------------------------------------

def _compare_plans(host, dev, refined=False):
    hp, hn, hex_ = host
    dp, dn, dex = dev
    assert np.array_equal(hn, dn), "vertex counts differ at %s" % np.nonzero(hn != dn)[0][:8]
    assert np.array_equal(hex_, dex), "the device expanded other cells than the host (same total order => same expansions)"
    for i in np.nonzero(hn > 0)[0]:
        assert np.array_equal(dp[i, :hn[i]], hp[i, :hn[i]]), (i, refined)   # bit for bit, createMoreVertexes' inserted vertices included


@pytest.mark.gpu
def test_device_map_and_path_search_equal_host_frontend():
    """fh_map_read == MapUtil::readMap restatement (occupancy cell for cell), fh_map_plan_batch == plan_path (vertices bit for bit,
    the same number of expanded cells) on the forest of BASELINE config 5; then the vertices Faster::replan would decompose."""
    from faster_amd import frontend

    n = 4096
    res, infl, zmax = 0.2, 0.3, 3.0
    cloud, cells, center, starts, goals = frontend.forest_queries(n, 21)
    hp, hn, hex_, hocc, hdims, horig = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, want_grid=True)
    m = capi.Map(0)
    m.read(cloud, cells, res, center, 0.0, zmax, infl)
    dims, orig = m.dims()
    assert np.array_equal(dims, hdims) and np.array_equal(orig, horig)
    assert np.array_equal(m.occupancy(), hocc)
    assert (hn > 0).mean() > 0.95 and hex_.mean() > 300
    _compare_plans((hp, hn, hex_), m.plan_batch(starts, goals))
    _compare_plans(frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8),
                   m.plan_batch(starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8), refined=True)
    # a second call on the same map object reuses the per-wavefront state (serial numbers instead of clearing)
    _compare_plans((hp, hn, hex_), m.plan_batch(starts, goals))
    # max_points too small is reported per query, not silently truncated
    _, dn_small, _ = m.plan_batch(starts[:256], goals[:256], max_points=3)
    assert np.array_equal(dn_small, np.where(hn[:256] > 3, -1, hn[:256]))
    m.close()


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["wall", "empty", "coarse", "tall", "large"])
def test_device_path_search_edge_cases(case):
    from faster_amd import frontend

    rng = np.random.default_rng(3)
    res, infl, zg, zmax = 0.25, 0.25, 0.0, 2.0
    cells, center = (40, 40, 8), np.array([5.0, 5.0, 1.0])
    if case == "wall":      # a full plane x = 5: nothing on the right is reachable from the left
        yy, zz = np.meshgrid(np.arange(-3, 13, 0.1), np.arange(-1, 3, 0.1))
        cloud = np.column_stack([np.full(yy.size, 5.0), yy.ravel(), zz.ravel()])
    elif case == "empty":
        cloud = np.zeros((0, 3))
    elif case == "coarse":  # inflation smaller than a cell (no cube), coarse cells
        cloud, _ = frontend.forest_cloud(4, size=(10.0, 10.0, 2.0), density=0.2)
        res, infl = 0.5, 0.2
        cells = (20, 20, 4)
    elif case == "tall":    # map centred high above the ground: z clipping of readMap on both sides
        cloud, _ = frontend.forest_cloud(5, size=(10.0, 10.0, 2.0), density=0.2)
        center = np.array([5.0, 5.0, 1.7])
        cells = (40, 40, 12)
    else:                   # FASTER's own resolution: 0.1 m cells, 115 x 115 x 20 = 264 500 of them, paths of ~100 cells
        cloud, _ = frontend.forest_cloud(8, size=(10.0, 10.0, 2.0), density=0.25)
        res, infl = 0.1, 0.3
        cells = (100, 100, 30)
    n = 256
    starts = np.column_stack([rng.uniform(0.5, 4.0, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
    goals = np.column_stack([rng.uniform(6.0, 9.5, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
    goals[:8] = starts[:8] + 0.01            # same cell: the path is [start, goal] (jps_manager.cpp:181-186)
    goals[8] = starts[8]                     # identical points
    starts[9] = [-50.0, 5.0, 1.0]            # outside the map: no path
    goals[10] = [5.0, 500.0, 1.0]
    starts[11, 2] = -0.7                     # z clamped to 0 (jps_manager.cpp:143-144)
    goals[12, 2] = -0.2
    starts[13] = goals[13] - [0.3, 0.0, 0.0]  # neighbouring cells
    host = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, want_grid=True)
    m = capi.Map(0)
    m.read(cloud, cells, res, center, zg, zmax, infl)
    assert np.array_equal(m.occupancy(), host[3])
    dev = m.plan_batch(starts, goals)
    _compare_plans(host[:3], dev)
    assert dev[1][9] == 0 and dev[1][10] == 0
    if case == "wall":
        assert (dev[1][16:] == 0).all(), "paths through a closed wall"
    if case == "empty":
        assert (dev[1][16:] == 2).all(), "free space: the cleaned path is the straight leg"
    _compare_plans(frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, max_points=24, max_vertex_dist=0.7, max_poly=0),
                   m.plan_batch(starts, goals, max_points=24, max_vertex_dist=0.7, max_poly=0), refined=True)
    m.close()


@pytest.mark.gpu
def test_device_corridor_front_end_equals_host(ctx):
    """The whole front-end of config C5 on the device (map -> path search -> createMoreVertexes/deleteVertexes -> decomposition -> rows in
    fh_problem's layout) against the CPU front-end: the same pairs kept and every polytope row BIT FOR BIT (both sides use only
    correctly rounded operations — no libm trigonometry, no fused multiply-adds — and the same tie rules), hence identical problems
    and identical solver results."""
    from faster_amd import frontend

    n = 3072
    hp, hf, hi = frontend.forest_batch(n, 31)
    vmap = capi.Map(0)
    dp, df, di = frontend.forest_batch(n, 31, front="device", ctx=ctx, vmap=vmap)
    vmap.close()
    assert np.array_equal(hi["kept"], di["kept"]) and len(hp) > 0.95 * n
    for f in abi.problem_dtype.names:
        assert np.array_equal(hp[f], dp[f]), f
    assert hf.shape == df.shape and np.array_equal(hf["a"], df["a"]) and np.array_equal(hf["b"], df["b"])
    assert di["front_timing"]["expansions"] > 300 * n


@pytest.mark.gpu
def test_results_do_not_depend_on_launch_order_or_publishing_ahead(ctx):
    """The scheduling devices of a big launch — hardest corridors first (a device-side sort), hard problems publishing frames ahead of
    the idle takers, work sharing itself — decide who explores what and when, never what comes out: 8192 fused pairs solved with each
    of them switched off in turn give the same result fields bit for bit."""
    import torch

    B, N = 8192, 10
    whole, faces, _ = corridor.whole_batch(B, seed=77, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    safe_t = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    dev = "cuda:0"
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    d_whole, d_faces = to_dev(whole), to_dev(faces)

    def run(c):
        d_safe, d_sf = to_dev(safe_t), torch.zeros_like(d_faces)
        d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
        d_sr = torch.zeros_like(d_wr)
        c.set_pair_margin(0.05)
        c.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(),
                             d_sr.data_ptr())
        c.sync()
        return d_wr.cpu().numpy().view(abi.result_dtype).copy(), d_sr.cpu().numpy().view(abi.result_dtype).copy()

    ref = run(ctx)
    assert (ref[1]["solved"] == 1).mean() > 0.7
    # (fh_set_sched: explicit fields, no environment.  min_nodes 64: almost nothing is given to idle workgroups; child_bound 0: every
    # child of a node is visited — the tree of the CPU oracle, about twice the nodes; workgroups_per_cu 5: also ANOTHER BUILD of the
    # kernel — compiled for two wavefronts per SIMD, nothing spilled — selected by any value up to 8)
    variants = {"launch_order": 0, "publish_factor": 0, "backlog": 0, "workgroups_per_cu": 5, "min_nodes": 64, "child_bound": 0, "look_every": 2,
                "look_every ": 64}
    for k, v in variants.items():
        ctx.set_sched(**{k.strip(): v})
        try:
            got = run(ctx)
        finally:
            ctx.set_sched()
        for a, b in zip(ref, got):
            for f in RESULT_FIELDS:
                assert np.array_equal(a[f], b[f]), (k, f)
        if k == "child_bound":  # the bound only removes nodes
            assert got[0]["nodes"].sum() > 1.3 * ref[0]["nodes"].sum() and got[1]["nodes"].sum() > ref[1]["nodes"].sum()
    solo = capi.Context(0)
    par = abi.default_params()
    par["share"] = 0
    solo.set_params(par)
    got = run(solo)
    solo.close()
    for a, b in zip(ref, got):
        for f in RESULT_FIELDS:
            assert np.array_equal(a[f], b[f]), ("share=0", f)


@pytest.mark.gpu
def test_independent_models_against_the_gpu_output_directly(ctx):
    """The independent legs — SciPy SLSQP on the reference's own unreduced 12N-coefficient model (oracle/py_model.py, written row by
    row as solverGurobi.cpp adds them), exhaustive enumeration of all P^N assignments with it, HiGHS on the mixed-integer constraint
    set — applied to what the GPU returned, not to the C oracle: the optimum of the GPU's assignment, its global optimality at the
    accepted dt, and the infeasibility of the factor before it."""
    from oracle import py_model

    def polys_of(p, faces):
        fb = int(p["face_begin"])
        return [(faces["a"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy(), faces["b"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy())
                for q in range(int(p["n_poly"]))]

    def args_of(p, faces):
        return (p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), bool(p["force_final_pos"]), polys_of(p, faces))

    # (1) the GPU's coefficients are the optimum of the unreduced model under the GPU's assignment
    pr, faces, _ = corridor.whole_batch(16, seed=203, n_seg=6, p_choices=(1, 2, 3))
    res = ctx.solve_batch(pr, faces)
    done = 0
    for i in np.nonzero(res["solved"])[0][:8]:
        p, r = pr[i], res[i]
        N = int(p["n_seg"])
        s = py_model.solve_fixed(N, float(r["dt"]), *args_of(p, faces), [int(a) for a in r["assign"][:N]])
        assert s is not None
        assert s[0] == pytest.approx(r["cost"], rel=1e-7, abs=1e-8)
        np.testing.assert_allclose(s[1], r["coeff"][:N], atol=5e-6)
        done += 1
    assert done >= 5
    # (2) no other assignment is better at the accepted dt (all P^N of them), and (3) the factor before the accepted one is infeasible
    pr, faces, _ = corridor.whole_batch(24, seed=204, n_seg=4, p_choices=(2, 3), speed=3.5, lateral=1.0)
    res = ctx.solve_batch(pr, faces)
    solved = np.nonzero(res["solved"])[0]
    assert len(solved) >= 6
    for i in solved[:4]:
        p, r = pr[i], res[i]
        best, barg, nfeas = py_model.enumerate_miqp(4, float(r["dt"]), *args_of(p, faces))
        assert nfeas >= 1 and best == pytest.approx(r["cost"], rel=1e-6, abs=1e-8), (i, best, r["cost"], barg, r["assign"][:4])
    later = [i for i in solved if res[i]["trials"] >= 2][:3]
    assert later, "no problem needed a second factor: make the corridors tighter"
    for i in later:
        p, r = pr[i], res[i]
        dt_before = float(r["dt"]) * (float(r["factor"]) - float(p["f_inc"])) / float(r["factor"])
        assert py_model.milp_feasible(4, dt_before, *args_of(p, faces), time_limit=30.0) is False, i


@pytest.mark.gpu
def test_device_path_search_against_pure_python_restatement():
    """fh_map_* against oracle/path_oracle.py directly (not through the C++ front-end): occupancy grid, vertices and expansion counts."""
    from faster_amd import frontend
    from oracle import path_oracle

    cloud, _ = frontend.forest_cloud(6, size=(8.0, 8.0, 2.0), density=0.2)
    res, infl, zg, zmax = 0.25, 0.25, 0.0, 2.0
    cells, center = (34, 34, 8), np.array([4.0, 4.0, 1.0])
    rng = np.random.default_rng(6)
    n = 16
    starts = np.column_stack([rng.uniform(0.5, 2.5, n), rng.uniform(0.5, 7.5, n), rng.uniform(0.3, 1.7, n)])
    goals = np.column_stack([rng.uniform(5.5, 7.5, n), rng.uniform(0.5, 7.5, n), rng.uniform(0.3, 1.7, n)])
    g = path_oracle.Grid(cloud.tolist(), cells, res, center.tolist(), zg, zmax, infl)
    m = capi.Map(0)
    m.read(cloud, cells, res, center, zg, zmax, infl)
    assert np.array_equal(m.occupancy().reshape(-1), np.frombuffer(bytes(g.occ), dtype=np.int8))
    dp, dn, dex = m.plan_batch(starts, goals)
    m.close()
    solved = 0
    for i in range(n):
        path, ex = path_oracle.plan(g, starts[i].tolist(), goals[i].tolist())
        if path is None:
            assert dn[i] == 0
            continue
        solved += 1
        assert dn[i] == len(path) and dex[i] == ex, (i, dn[i], len(path), dex[i], ex)
        assert np.array_equal(dp[i, :dn[i]], np.array(path)), i
    assert solved >= 12
