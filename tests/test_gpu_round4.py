"""GPU tests added in round 4: the device pipeline of one Faster::replan per pair END TO END against the restatement of the caller
(faster_amd/host/replan_stub.hpp = faster/src/faster.cpp:296-595), with unknown space given as an INPUT (fh_set_unknown_grid_device,
fh_pair_rule mode 2) — the same unknown voxels that the stub receives as a point cloud (updateMap, faster.cpp:99-137)."""
import os
import subprocess

import numpy as np
import pytest

from faster_amd import abi, capi, corridor, frontend

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _torch_first():
    import torch  # noqa: F401  (torch before the HIP library: one HIP runtime in the process)


def _build_pairs_driver():
    from faster_amd import build as fb

    fb.build_all()
    exe = os.path.join(ROOT, "tests", "cpp", "test_replan_pairs")
    src = exe + ".cpp"
    host = os.path.join(ROOT, "faster_amd", "host")
    deps = [src, fb.HOST_SO] + [os.path.join(host, f) for f in ("replan_stub.hpp", "corridor_frontend.hpp", "corridor_frontend.cpp", "solver_hip.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-I", os.path.join(ROOT, "include"), "-I", host, src, os.path.join(host, "corridor_frontend.cpp"),
                               "-o", exe, "-L", os.path.join(ROOT, "faster_amd"), "-lsolverhip", "-lfasterhip", "-ldl", "-fopenmp",
                               "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    return exe


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def device_replans(cloud, cells, center, starts, vels, goals, flags, origin, dims, P):
    """The stages of bench.py's `replan_faithful`, unknown space as an input (rule mode 2).  Returns what the device committed."""
    import torch

    dev = "cuda:0"
    B, N, max_poly, fpp, max_states = len(starts), P["N"], P["max_poly"], 192, 1024  # (3 polytopes of up to 64 rows: what SolverHip accepts)
    mp = 16   # vertices of JPS_in: the WHOLE path inside the sphere (the reference cuts a copy to max_poly legs for the whole corridor only)
    whole = abi.make_problems(B)
    whole["n_seg"], whole["force_final_pos"], whole["dc"] = N, 1, P["dc"]
    whole["v_max"], whole["a_max"], whole["j_max"] = P["v_max"], P["a_max"], P["j_max"]
    whole["f_init"], whole["f_final"], whole["f_inc"] = 1.0, 10.0, 1.0
    whole["x0"][:, 0:3], whole["x0"][:, 3:6] = starts, vels
    tmpl = corridor.safe_templates(whole)
    tmpl["n_seg"] = N
    ctx, vmap = capi.Context(0), capi.Map(0)
    try:
        f64, i32 = torch.float64, torch.int32
        d_cloud, d_starts, d_goals, d_flags = _dev(cloud), _dev(starts), _dev(goals), _dev(flags)
        d_whole, d_safe = _dev(whole), _dev(tmpl)
        d_paths, d_np, d_ex = torch.zeros((B, mp, 3), dtype=f64, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=torch.int64, device=dev)
        FB, RES = abi.face_dtype.itemsize, abi.result_dtype.itemsize
        d_wf, d_sf = torch.zeros(B * fpp * FB, dtype=torch.uint8, device=dev), torch.zeros(B * fpp * FB, dtype=torch.uint8, device=dev)
        d_off, d_npoly, d_last = torch.zeros((B, 9), dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros((B, 3), dtype=f64, device=dev)
        d_wr, d_sr = torch.zeros(B * RES, dtype=torch.uint8, device=dev), torch.zeros(B * RES, dtype=torch.uint8, device=dev)
        d_plans = torch.zeros(B * max_states * abi.state_dtype.itemsize, dtype=torch.uint8, device=dev)
        d_counts, d_k = torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev)
        d_spaths, d_snp = torch.zeros((B, max_poly + 1, 3), dtype=f64, device=dev), torch.zeros(B, dtype=i32, device=dev)
        ctx.set_pair_rule(mode=2, drone_radius=P["drone_radius"], delta_h=P["delta_h"], delta_a=P["delta_a"])
        ctx.set_unknown_grid_device(d_flags.data_ptr(), origin, P["res"], dims)
        vmap.set_search("jps")
        vmap.set_sphere(P["Ra"])
        vmap.read_device(d_cloud.data_ptr(), len(cloud), cells, P["res"], center, 0.0, P["z_max"], P["inflation"])
        vmap.plan_batch_device(d_starts.data_ptr(), d_goals.data_ptr(), B, mp, d_paths.data_ptr(), d_np.data_ptr(), d_ex.data_ptr(), P["dist_max_vertexes"], 0)
        vmap.sync()
        ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), d_paths.data_ptr(), d_np.data_ptr(), B, mp, max_poly, fpp, d_wf.data_ptr(), d_off.data_ptr(),
                                  d_npoly.data_ptr(), d_last.data_ptr(), P["decomp_radius"], 0.0)
        ctx.corridor_problems_device(d_np.data_ptr(), d_last.data_ptr(), d_goals.data_ptr(), d_wf.data_ptr(), d_off.data_ptr(), d_npoly.data_ptr(), B, fpp, N,
                                     d_whole.data_ptr())
        ctx.solve_batch_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, d_wr.data_ptr())
        ctx.safe_corridor_batch_device(d_whole.data_ptr(), d_wr.data_ptr(), d_paths.data_ptr(), d_np.data_ptr(), mp, d_goals.data_ptr(), d_cloud.data_ptr(),
                                       len(cloud), origin, P["res"], dims, B, 0.5, max_poly, (2.0, 2.0, 1.0), P["decomp_radius"], 0.0, fpp, N,
                                       d_safe.data_ptr(), d_sf.data_ptr(), d_spaths.data_ptr(), d_snp.data_ptr())
        ctx.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, fpp, d_sr.data_ptr())
        ctx.append_plans_device(d_whole.data_ptr(), d_wr.data_ptr(), d_safe.data_ptr(), d_sr.data_ptr(), B, 0.5, max_states, d_plans.data_ptr(),
                                d_counts.data_ptr(), d_k.data_ptr())
        ctx.sync()
        # the consumer of the plans: Faster::getNextGoal for every pair, 1 call, then 7 more, then far beyond the end of every plan
        d_cursor = torch.zeros(B, dtype=i32, device=dev)
        d_goal_states = torch.zeros(B * abi.state_dtype.itemsize, dtype=torch.uint8, device=dev)
        d_okg = torch.zeros(B, dtype=i32, device=dev)
        next_goals = []
        for ticks in (1, 7, 5000):
            ctx.next_goals_device(d_plans.data_ptr(), d_counts.data_ptr(), d_cursor.data_ptr(), B, max_states, ticks, d_goal_states.data_ptr(), d_okg.data_ptr())
            ctx.sync()
            next_goals.append((d_goal_states.cpu().numpy().view(abi.state_dtype).copy(), d_cursor.cpu().numpy().copy(), d_okg.cpu().numpy().copy()))
        out = {"next_goals": next_goals, "np": d_np.cpu().numpy(), "wres": d_wr.cpu().numpy().view(abi.result_dtype).copy(), "sres": d_sr.cpu().numpy().view(abi.result_dtype).copy(),
               "safe": d_safe.cpu().numpy().view(abi.problem_dtype).copy(), "whole": d_whole.cpu().numpy().view(abi.problem_dtype).copy(),
               "counts": d_counts.cpu().numpy(), "k": d_k.cpu().numpy(),
               "plans": d_plans.cpu().numpy().view(abi.state_dtype).reshape(B, max_states).copy(), "snp": d_snp.cpu().numpy(),
               "spaths": d_spaths.cpu().numpy(), "paths": d_paths.cpu().numpy(), "sfaces": d_sf.cpu().numpy().view(abi.face_dtype).reshape(B, fpp).copy()}
        # rule mode 2 without a grid is refused loudly, by the staged entry points and by the fused pair kernel
        d_wr2, d_sr2, d_sf2, d_safe2 = torch.zeros_like(d_wr), torch.zeros_like(d_sr), torch.zeros_like(d_sf), _dev(tmpl)
        ctx.set_unknown_grid_device(None)
        with pytest.raises(capi.FasterHipError):
            ctx.append_plans_device(d_whole.data_ptr(), d_wr.data_ptr(), d_safe.data_ptr(), d_sr.data_ptr(), B, 0.5, max_states, d_plans.data_ptr(),
                                    d_counts.data_ptr(), d_k.data_ptr())
        with pytest.raises(capi.FasterHipError):
            ctx.solve_pairs_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, 0.5, 0.0, 3, d_wr2.data_ptr(), d_safe2.data_ptr(), d_sf2.data_ptr(), d_sr2.data_ptr())
        # [r5] the FUSED pair kernel with the same unknown voxels (rule mode 2 inside the one launch: whole solve -> H, R, "is a safe
        # trajectory needed" as the reference decides them -> safe solve in the polytopes of the whole corridor from the one that holds R)
        ctx.set_unknown_grid_device(d_flags.data_ptr(), origin, P["res"], dims)
        ctx.set_pair_margin(0.0)
        ctx.solve_pairs_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, 0.5, 0.0, max_poly, d_wr2.data_ptr(), d_safe2.data_ptr(), d_sf2.data_ptr(), d_sr2.data_ptr())
        ctx.sync()
        ctx.set_pair_margin(-1.0)
        out.update({"fused_wres": d_wr2.cpu().numpy().view(abi.result_dtype).copy(), "fused_sres": d_sr2.cpu().numpy().view(abi.result_dtype).copy(),
                    "fused_safe": d_safe2.cpu().numpy().view(abi.problem_dtype).copy(), "fused_sfaces": d_sf2.cpu().numpy().view(abi.face_dtype).copy(),
                    "fused_kernel": ctx.last_launch()[1]})
        dm, og = vmap.dims()
    finally:
        vmap.close()
        ctx.close()
    return out


def stub_replans(tmp_path, cloud, cells, center, starts, vels, goals, unknown_pts, P):
    """The same pairs, one at a time, through replan_stub.hpp driving SolverHip (tests/cpp/test_replan_pairs.cpp)."""
    exe = _build_pairs_driver()
    B = len(starts)
    hi = np.zeros(16, dtype=np.int32)
    hi[:10] = [P["N"], P["N"], P["max_poly"], P["max_poly"], cells[0], cells[1], cells[2], B, len(cloud), len(unknown_pts)]
    hd = np.zeros(32, dtype=np.float64)
    hd[:17] = [P["dc"], P["v_max"], P["a_max"], P["j_max"], P["Ra"], P["drone_radius"], P["decomp_radius"], P["dist_max_vertexes"], P["delta_a"],
               P["delta_h"], P["res"], P["inflation"], 0.0, P["z_max"], center[0], center[1], center[2]]
    sc, outp = tmp_path / "pairs.bin", tmp_path / "pairs.out"
    with open(sc, "wb") as f:
        f.write(hi.tobytes())
        f.write(hd.tobytes())
        f.write(np.ascontiguousarray(cloud, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(unknown_pts, dtype=np.float64).tobytes())
        f.write(np.ascontiguousarray(np.concatenate([starts, vels, goals], axis=1), dtype=np.float64).tobytes())
    r = subprocess.run([exe, str(sc), str(outp)], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    if os.environ.get("FH_DEBUG_PAIR"):
        print(r.stderr[-3000:])
    raw = open(outp, "rb").read()
    recs, pos = [], 0
    for _ in range(B):
        rec = np.frombuffer(raw, dtype=np.int32, count=8, offset=pos)
        fac = np.frombuffer(raw, dtype=np.float64, count=2, offset=pos + 32)
        pos += 48
        cnt = int(rec[7])
        plan = np.frombuffer(raw, dtype=np.float64, count=12 * cnt, offset=pos).reshape(cnt, 12)
        pos += 96 * cnt
        recs.append({"ok": int(rec[0]), "stage": int(rec[1]), "needed_safe": int(rec[2]), "index_H": int(rec[3]), "k_safe": int(rec[4]),
                     "n_whole": int(rec[5]), "n_safe": int(rec[6]), "plan": plan, "whole_factor": fac[0], "safe_factor": fac[1]})
    assert pos == len(raw)
    return recs


def test_device_replan_chain_equals_the_stub_with_unknown_space_as_an_input(tmp_path):
    """1024 independent start/goal pairs in a random forest of which only some regions have been seen.  Device: map -> jump point search
    (JPS_in) -> whole corridor -> whole solve -> H, R and the safe corridor against the UNKNOWN VOXELS GIVEN AS AN INPUT -> safe solve ->
    appendToPlan, every stage a batch launch.  Caller: replan_stub.hpp, one pair at a time, the unknown voxels as the point cloud a
    mapper would hand over (z-major), brute-force nearest-neighbour queries where the reference asks its kd-tree.  Per pair: the same
    decision at every stage (path found, whole trajectory, safe trajectory needed, safe trajectory found), the same sample k_safe, the
    same number of committed states, the committed states themselves to 1e-9."""
    B = 1024
    P = {"N": 6, "max_poly": 3, "dc": 0.01, "v_max": 5.0, "a_max": 5.0, "j_max": 8.0, "Ra": 4.0, "drone_radius": 0.3, "decomp_radius": 0.05,
         "dist_max_vertexes": 1.5, "delta_a": 0.5, "delta_h": 1.0, "res": 0.2, "inflation": 0.3, "z_max": 3.0}
    cloud, cells, center, starts, goals, rng = frontend.forest_queries(B, 11, return_rng=True)
    u = goals - starts
    u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-9)
    vels = u * rng.uniform(0, 1.5, size=(B, 1))
    # the lattice of the device map (MapUtil::readMap: dims include the inflation margin) carries the unknown flags
    probe = capi.Map(0)
    probe.read(cloud, cells, P["res"], center, 0.0, P["z_max"], P["inflation"])
    dims, origin = probe.dims()
    probe.close()
    dims, origin = [int(d) for d in dims], np.array(origin, dtype=np.float64)
    iz, iy, ix = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
    centres = np.stack([(ix + 0.5) * P["res"] + origin[0], (iy + 0.5) * P["res"] + origin[1], (iz + 0.5) * P["res"] + origin[2]], axis=-1)
    seen = np.zeros(iz.shape, dtype=bool)
    for c in rng.uniform([1, 1, 1.5], [19, 19, 1.5], size=(16, 3)):      # what the vehicle has explored: a few overlapping spheres
        seen |= np.linalg.norm(centres - c, axis=-1) < rng.uniform(2.0, 3.5)
    flags = (~seen).astype(np.uint8)                                      # [nz][ny][nx], x fastest
    unknown_pts = centres[~seen]                                          # z-major, x fastest: the order the device enumerates
    assert 0.2 < flags.mean() < 0.9

    dv = device_replans(cloud, cells, center, starts, vels, goals, flags, origin, dims, P)
    st = stub_replans(tmp_path, cloud, cells, center, starts, vels, goals, unknown_pts, P)

    # the safe corridors of a sample of pairs: the device decomposition against the unknown voxels of the grid + the occupied points
    # equals the host decomposition of the same path against the explicit cloud [unknown voxels | occupied], row for row
    both = np.concatenate([unknown_pts, cloud])
    checked = 0
    for i in np.nonzero(dv["safe"]["n_seg"] > 0)[0][:48]:
        k = int(dv["snp"][i])
        polys, _ = frontend.decompose(dv["spaths"][i, :k], both, P["decomp_radius"], 0.0)
        off = dv["safe"]["face_off"][i]
        assert dv["safe"]["n_poly"][i] == len(polys) == k - 1, i
        for p_, (A, b) in enumerate(polys):
            rows = dv["sfaces"][i, off[p_]:off[p_ + 1]]
            assert len(rows) == len(b), (i, p_, len(rows), len(b))
            assert np.array_equal(rows["a"], A) and np.array_equal(rows["b"], b), (i, p_)
        checked += 1
    assert checked >= 32
    # ... and the safe paths themselves equal the numpy restatement of the march against the same unknown points (oracle/pair_glue.py)
    from oracle import pair_glue

    for i in list(np.nonzero(dv["snp"] >= 2)[0][:64]) + ([566] if B > 566 and dv["snp"][566] >= 2 else []):
        want = pair_glue.safe_path(dv["paths"][i, : dv["np"][i]], starts[i], dv["safe"]["x0"][i, :3], 0.0, P["drone_radius"], P["max_poly"],
                                   unknown_pts=unknown_pts)
        got = dv["spaths"][i, : dv["snp"][i]]
        if len(want) != len(got) or np.abs(want - got).max() > 1e-9:
            print("safe path of pair", i, "device", got.tolist(), "numpy", want.tolist(), "JPS_in", dv["paths"][i, : dv["np"][i]].tolist())
        assert len(want) == len(got) and np.abs(want - got).max() <= 1e-9, i
    whole_ok = (dv["np"] >= 2) & (dv["wres"]["solved"] == 1) & (dv["whole"]["n_seg"] > 0)
    need_safe = dv["safe"]["n_seg"] > 0
    safe_ok = dv["sres"]["solved"] == 1
    stages = {1: 0, 2: 0, 3: 0, 5: 0}
    worst = 0.0
    for i, s in enumerate(st):
        stages[s["stage"]] = stages.get(s["stage"], 0) + 1
        if s["stage"] == 1:                       # no path
            assert dv["np"][i] < 2 and dv["counts"][i] == 0, i
            continue
        if s["stage"] == 2:                       # no whole trajectory
            assert not whole_ok[i] and dv["counts"][i] == 0, i
            continue
        assert whole_ok[i], i
        assert s["n_whole"] == max(2, int(P["N"] * dv["wres"]["dt"][i] / P["dc"])), i
        assert s["whole_factor"] == dv["wres"]["factor"][i], i
        assert bool(s["needed_safe"]) == bool(need_safe[i] or (dv["snp"][i] >= 2 and not need_safe[i])), i
        if s["stage"] == 3:                       # a safe trajectory was needed and not found
            assert not (need_safe[i] and safe_ok[i]) and dv["counts"][i] == 0, i
            continue
        assert s["stage"] == 5 and s["ok"] == 1, (i, s["stage"])
        assert dv["k"][i] == s["k_safe"], (i, dv["k"][i], s["k_safe"], s["index_H"], "device: safe n_seg %d n_poly %d solved %d status %d trials %d, "
                                           "safe path %d vertices %s, x0 %s xf %s | stub: needed_safe %d n_safe %d safe_factor %g" % (
                                               dv["safe"]["n_seg"][i], dv["safe"]["n_poly"][i], dv["sres"]["solved"][i], dv["sres"]["status"][i],
                                               dv["sres"]["trials"][i], dv["snp"][i], dv["spaths"][i].tolist(), dv["safe"]["x0"][i].tolist(),
                                               dv["safe"]["xf"][i].tolist(), s["needed_safe"], s["n_safe"], s["safe_factor"]))
        if s["needed_safe"]:
            if not (need_safe[i] and safe_ok[i] and s["safe_factor"] == dv["sres"]["factor"][i]):
                print("pair", i, "device: need_safe", bool(need_safe[i]), "solved", bool(safe_ok[i]), "factor", float(dv["sres"]["factor"][i]), "dt",
                      repr(float(dv["sres"]["dt"][i])), "status", int(dv["sres"]["status"][i]), "trials", int(dv["sres"]["trials"][i]), "rows",
                      np.diff(dv["safe"]["face_off"][i][: dv["safe"]["n_poly"][i] + 1]).tolist(), "safe path",
                      [tuple(float(x) for x in v) for v in dv["spaths"][i, : dv["snp"][i]]], "x0", dv["safe"]["x0"][i].tolist(), "xf",
                      dv["safe"]["xf"][i].tolist(), "| stub factor", s["safe_factor"], "n_safe", s["n_safe"])
            assert need_safe[i] and safe_ok[i] and s["safe_factor"] == dv["sres"]["factor"][i], i
        n = len(s["plan"])
        assert dv["counts"][i] == n, (i, dv["counts"][i], n)
        got = dv["plans"][i, :n]
        mine = np.concatenate([got["pos"], got["vel"], got["accel"], got["jerk"]], axis=1)
        worst = max(worst, float(np.abs(mine - s["plan"]).max()))
    assert worst < 1e-9, worst
    # [r5] the fused pair kernel with unknown space as an input: the whole results of the staged launch bit for bit; per pair the
    # reference's decisions as the caller's restatement takes them — a safe trajectory is needed (H exists) exactly where the stub says so,
    # and R (the state that becomes x0 of the safe problem: pos, vel, accel) is the staged chain's R (to 1e-12), which the committed
    # plans above tie to the stub's k_safe — and the safe solves equal the oracle on the problems the kernel wrote
    from oracle import oracle as orc

    fw, fs, fsafe = dv["fused_wres"], dv["fused_sres"], dv["fused_safe"]
    assert dv["fused_kernel"] == "fh::solve_kernel<6, true, 2, true>", dv["fused_kernel"]
    for f in (n for n in abi.result_dtype.names if n not in ("nodes", "qp_iters", "kflops")):
        assert np.array_equal(fw[f], dv["wres"][f]), f
    fused_need = fsafe["n_seg"] > 0
    n_need = n_same_r = 0
    for i, s in enumerate(st):
        if s["stage"] in (1, 2):
            assert not fused_need[i], i                       # no whole trajectory: nothing to hand over
            continue
        if not s["needed_safe"]:
            assert not fused_need[i], (i, "the stub needs no safe trajectory here")
            continue
        n_need += 1
        if need_safe[i]:                                      # (the staged chain wrote a safe problem: its x0 is R)
            assert fused_need[i], (i, "the stub needs a safe trajectory here")
            # (1e-12: another sample k would move R by a whole step of the trajectory; the two kernels may contract multiply-adds differently)
            assert np.allclose(fsafe["x0"][i], dv["safe"]["x0"][i], rtol=0, atol=1e-12), (i, fsafe["x0"][i], dv["safe"]["x0"][i])
            n_same_r += 1
    assert n_need > 0.15 * B and n_same_r > 0.9 * n_need, (n_need, n_same_r)
    orc.build()
    live = np.nonzero(fused_need)[0]
    ref = orc.solve_batch(fsafe[live], dv["fused_sfaces"])
    for f in ("solved", "trials", "factor", "dt", "status"):
        assert np.array_equal(fs[f][live], ref[f]), f
    oks = ref["solved"] == 1
    np.testing.assert_allclose(fs["cost"][live][oks], ref["cost"][oks], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(fs["coeff"][live][oks], ref["coeff"][oks], rtol=0, atol=1e-6)
    print("fused pair kernel, rule mode 2: %d pairs need a safe trajectory, R identical in %d, %d safe problems solved" % (n_need, n_same_r, int(oks.sum())))
    # getNextGoal on the committed plans (faster.cpp:699-723: front(), pop_front() while more than one state is left): calls 1, 8 and 5008
    called = 0
    for (g, cur, okg), ticks in zip(dv["next_goals"], (1, 7, 5000)):
        called += ticks
        for i in range(B):
            n = int(dv["counts"][i])
            if n == 0:
                assert okg[i] == 0 and not g["pos"][i].any() and not g["vel"][i].any(), i
                continue
            want_front = min(called - 1, n - 1)          # the state the last of the calls so far returned
            assert okg[i] == 1 and cur[i] == min(called, n - 1), (i, cur[i], called, n)
            assert np.array_equal(g["pos"][i], dv["plans"]["pos"][i, want_front]) and np.array_equal(g["jerk"][i], dv["plans"]["jerk"][i, want_front]), i
    # the batch exercises every outcome that matters
    assert stages[5] > 0.5 * B and sum(1 for s in st if s["needed_safe"]) > 0.15 * B and sum(1 for s in st if s["stage"] == 5 and not s["needed_safe"]) > 10
    print("replan chain == stub on %d pairs: stages %s, worst state difference %.2e" % (B, stages, worst))


def test_hashed_cell_records_give_the_same_paths_in_a_fraction_of_the_workspace():
    """fh_map_set_records: the jump point search with a hashed table of the cells a query has reached instead of one record per
    cell of the map and wavefront.  Same popped nodes and vertices bit for bit as the host restatement (plan_path_jps, pinned to
    the reference's compiled jps3d) on the forest of config C5 and on a 0.1 m map; repeated calls (stale slots of earlier
    queries count as free); a table that is too small for some queries (1024 slots: chains collide, lanes contend for slots)
    answers -2 for exactly the queries that reach more than 768 cells and is identical on the others; the workspace no longer
    grows with the map."""
    frontend.set_search("jps")
    m = capi.Map(0)
    try:
        m.set_search("jps")
        res, infl, zmax = 0.2, 0.3, 3.0
        cloud, cells, center, starts, goals = frontend.forest_queries(4096, 23)
        host = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals)
        m.read(cloud, cells, res, center, 0.0, zmax, infl)
        m.set_records(0)
        dense = m.plan_batch(starts, goals)
        dense_bytes = m.workspace_bytes()

        def same(a, b, what, skip=None):
            keep = np.ones(len(a[1]), bool) if skip is None else ~skip
            assert np.array_equal(a[1][keep], b[1][keep]), what
            assert np.array_equal(a[2][keep], b[2][keep]), (what, "other nodes popped")
            for i in np.nonzero(keep & (a[1] > 0))[0]:
                assert np.array_equal(a[0][i, :a[1][i]], b[0][i, :a[1][i]]), (what, i)

        same(host, dense, "per-cell records")
        m.set_records(16384)
        for rep in range(3):
            same(host, m.plan_batch(starts, goals), "hashed records, call %d" % rep)
        hashed_bytes = m.workspace_bytes()
        assert hashed_bytes < dense_bytes / 3 and hashed_bytes <= 5120 * 16384 * 40
        refined = (16, 1.5, 8)
        same(frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=refined[0], max_vertex_dist=refined[1], max_poly=refined[2]),
             m.plan_batch(starts, goals, max_points=refined[0], max_vertex_dist=refined[1], max_poly=refined[2]), "hashed records, refined vertices")
        # a table that overflows for the long searches
        m.set_records(1024)
        small = m.plan_batch(starts, goals)
        over = small[1] == -2
        assert 0 < over.sum() < len(over) // 2, over.sum()
        same(host, small, "1024 slots", skip=over)
        assert host[2][over].min() > 200  # (only long searches overflow: a search reaches a few cells per popped node)
        m.set_records(-1)  # (by the size of the map: per-cell records here)
        same(host, m.plan_batch(starts, goals), "back to per-cell records")
        assert m.workspace_bytes() == dense_bytes
        # a 0.1 m map
        rng = np.random.default_rng(9)
        cloud, _ = frontend.forest_cloud(8, size=(10.0, 10.0, 2.0), density=0.25)
        res, infl, cells, center = 0.1, 0.3, (100, 100, 30), np.array([5.0, 5.0, 1.0])
        n = 512
        starts = np.column_stack([rng.uniform(0.5, 4.0, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
        goals = np.column_stack([rng.uniform(6.0, 9.5, n), rng.uniform(0.5, 9.5, n), rng.uniform(0.3, 1.7, n)])
        goals[:8] = starts[:8] + 0.01
        goals[8] = starts[8]
        starts[9] = [-50.0, 5.0, 1.0]
        m.read(cloud, cells, res, center, 0.0, 2.0, infl)
        host = frontend.plan_batch(cloud, cells, res, center, 0.0, 2.0, infl, starts, goals)
        m.set_records(32768)
        same(host, m.plan_batch(starts, goals), "0.1 m cells, hashed records")
        assert m.workspace_bytes() <= 5120 * 32768 * 40
        # by the size of the map (-1): per-cell records while the budget holds them for half of the wavefronts, hashed beyond
        m.set_records(-1)
        same(host, m.plan_batch(starts, goals), "0.1 m cells, records by the size of the map")
        assert m.workspace_bytes() > 5120 * 300000 * 16  # (300 000 cells: per-cell records)
        cloud, cells, center, starts, goals = frontend.forest_queries(128, 5)
        cells = tuple(2 * c for c in cells)  # the C5 forest at 0.1 m: 1.45 M cells
        m.read(cloud, cells, 0.1, center, 0.0, 3.0, infl)
        host = frontend.plan_batch(cloud, cells, 0.1, center, 0.0, 3.0, infl, starts, goals)
        same(host, m.plan_batch(starts, goals), "1.45 M cells, records by the size of the map")
        assert m.workspace_bytes() <= 5120 * 131072 * 40  # (hashed: 27 GB, not the 48 GB of as many per-cell records as fit)
    finally:
        m.close()
        frontend.set_search("astar")


def test_jump_point_search_with_heaps_far_deeper_than_lds():
    """A cube of 81 x 81 x 80 cells with dense trees and goals that cannot be reached from many starts: the searches exhaust what
    they can reach (tens of thousands of popped nodes, heaps of tens of thousands of entries — the 311 entries of LDS are the tip),
    so that the sinking of an entry continues in the chunk pool, entries rise across the boundary and `increase` finds its entry in
    the pool.  Same popped nodes and vertices as the host restatement, with per-cell records and with a hashed table big enough."""
    frontend.set_search("jps")
    m = capi.Map(0)
    try:
        m.set_search("jps")
        side, res, infl = 12.0, 0.15, 0.45
        cloud, _ = frontend.forest_cloud(116, size=(side, side, side), density=0.3)
        cells = (int(side / res) + 1, int(side / res) + 1, int(side / res))
        center = np.array([side / 2, side / 2, side / 2])
        rng = np.random.default_rng(3)
        n = 192
        starts = np.column_stack([rng.uniform(0.3, side - 0.3, n), rng.uniform(0.3, side - 0.3, n), rng.uniform(0.0, side, n)])
        goals = np.column_stack([rng.uniform(0.3, side - 0.3, n), rng.uniform(0.3, side - 0.3, n), rng.uniform(0.0, side, n)])
        host = frontend.plan_batch(cloud, cells, res, center, 0.0, side, infl, starts, goals)
        assert host[2].max() > 30000 and (host[1] == 0).sum() >= 4 and (host[1] > 0).sum() >= 100, (host[2].max(), (host[1] == 0).sum())
        m.read(cloud, cells, res, center, 0.0, side, infl)
        for slots in (0, 262144):
            m.set_records(slots)
            dp, dn, dex = m.plan_batch(starts, goals)
            assert np.array_equal(host[1], dn), (slots, np.nonzero(host[1] != dn)[0][:8], dn[host[1] != dn][:8])
            assert np.array_equal(host[2], dex), slots
            for i in np.nonzero(dn > 0)[0]:
                assert np.array_equal(host[0][i, :dn[i]], dp[i, :dn[i]]), (slots, i)
    finally:
        m.close()
        frontend.set_search("astar")


def test_the_two_builds_of_the_solve_kernel_give_the_same_bits():
    """fh_sched.workgroups_per_cu <= 8 selects solve_kernel<N, PAIRS, 2> — the same source compiled for two wavefronts per SIMD (all
    registers, no scratch; a batch alone on the device is done sooner; small batches run it by default): plain batches at every
    instantiation of N give the same result fields bit for bit as the throughput build, which a value above 8 asks for (the fused
    pairs: test_results_do_not_depend_on_launch_order_or_publishing_ahead; the oracle parity of both: tests/tools/parity_sweep.py
    alternates between them)."""
    fields = ("solved", "status", "trials", "factor", "dt", "cost", "coeff", "assign")  # (not the work counters: who explores what may differ)
    c = capi.Context(0)
    try:
        for n_seg, pch, n in ((5, (1, 2, 3), 2048), (10, (3, 4, 5, 6), 2048), (13, (4, 5), 512), (16, (3, 4), 256)):
            pr, faces, _ = corridor.make_batch(n, n_seg, pch, True, 1000 + n_seg)
            c.set_sched(workgroups_per_cu=12)  # (above 8: the three-wavefront build whatever the batch)
            ref = c.solve_batch(pr, faces)
            assert (ref["solved"] == 1).mean() > 0.3
            for wpc in (8, 0):  # (8: the two-wavefront build; 0: the library's choice — that build again for a batch this small)
                c.set_sched(workgroups_per_cu=wpc)
                got = c.solve_batch(pr, faces)
                for f in fields:
                    assert np.array_equal(ref[f], got[f]), (n_seg, wpc, f)
    finally:
        c.close()


def test_the_two_builds_of_the_pair_kernel_give_the_same_bits():
    """The same for the fused whole -> hand-off -> safe kernel at the instantiations the big test does not reach (N = 6: FASTER's own
    N_whole = N_safe; N = 15: config C5): fh_sched.workgroups_per_cu 12 (three wavefronts per SIMD) against 8 (two)."""
    import torch

    fields = [n for n in abi.result_dtype.names if n not in ("nodes", "qp_iters", "kflops")]
    dev = "cuda:0"
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    c = capi.Context(0)
    try:
        for N, pch, B in ((6, (2, 3), 1024), (15, (4, 5, 6), 256)):
            whole, faces, _ = corridor.whole_batch(B, seed=500 + N, n_seg=N, p_choices=pch)
            safe_t = corridor.safe_templates(whole)
            mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
            d_whole, d_faces = to_dev(whole), to_dev(faces)
            out = []
            for wpc in (12, 8):
                c.set_sched(workgroups_per_cu=wpc)
                c.set_pair_margin(0.05)
                d_safe, d_sf = to_dev(safe_t), torch.zeros_like(d_faces)
                d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
                d_sr = torch.zeros_like(d_wr)
                c.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(),
                                     d_sf.data_ptr(), d_sr.data_ptr())
                c.sync()
                out.append((d_wr.cpu().numpy().view(abi.result_dtype).copy(), d_sr.cpu().numpy().view(abi.result_dtype).copy(),
                            d_safe.cpu().numpy().copy(), d_sf.cpu().numpy().copy()))
            assert (out[0][0]["solved"] == 1).mean() > 0.5 and (out[0][1]["solved"] == 1).mean() > 0.3
            for k in (0, 1):
                for f in fields:
                    assert np.array_equal(out[0][k][f], out[1][k][f]), (N, k, f)
            assert np.array_equal(out[0][2], out[1][2]) and np.array_equal(out[0][3], out[1][3]), N  # the safe problems and their faces
    finally:
        c.close()
