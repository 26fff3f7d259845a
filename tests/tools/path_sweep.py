"""Randomized parity sweep of the device front-end (fh_map_* + fh_corridor_batch_device) against the CPU front-end: maps of different
size, resolution, inflation and tree density; per configuration the occupancy grid, every path vertex, every expansion count and every
polytope row are compared bit for bit.  usage (GPU box, PYTHONPATH = repo root): path_sweep.py [queries_per_config] [configs] [astar|jps]
[record slots] (jps: jump point search in jps3d's own order on both sides — plan_path_jps vs fh_map_set_search(1); maps as high as wide
included; record slots: fh_map_set_records, e.g. 16384 for the hashed cell records)."""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

from faster_amd import abi, capi, frontend

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ncfg = int(sys.argv[2]) if len(sys.argv) > 2 else 12
search = sys.argv[3] if len(sys.argv) > 3 else "astar"
slots = int(sys.argv[4]) if len(sys.argv) > 4 else -1
rng = np.random.default_rng(2024)
ctx, vmap = capi.Context(0), capi.Map(0)
frontend.set_search(search)
vmap.set_search(search)
vmap.set_records(slots)
tot_q = tot_exp = bad = n_limit = n_limit_dev = 0
t0 = time.time()
for c in range(ncfg):
    side = float(rng.choice([8.0, 12.0, 20.0]))
    height = float(rng.choice([2.0, 3.0])) if (search == "astar" or c % 4) else side   # (jps: every fourth map is a cube)
    res = float(rng.choice([0.15, 0.2, 0.25, 0.3]))
    infl = float(rng.choice([0.0, 0.2, 0.3, 0.45]))
    dens = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
    mvd = float(rng.choice([0.8, 1.5, 2.5]))
    max_poly = int(rng.choice([4, 6, 8]))
    size = (side, side, height)
    cloud, _ = frontend.forest_cloud(100 + c, size=size, density=dens)
    cells = (int(side / res) + 6, int(side / res) + 6, int(height / res))
    center = np.array([side / 2, side / 2, height / 2])
    starts = np.column_stack([rng.uniform(0.3, side - 0.3, nq), rng.uniform(0.3, side - 0.3, nq), rng.uniform(-0.2, height, nq)])
    goals = np.column_stack([rng.uniform(0.3, side - 0.3, nq), rng.uniform(0.3, side - 0.3, nq), rng.uniform(-0.2, height, nq)])
    if os.environ.get("ONLY_CFG") and int(os.environ["ONLY_CFG"]) != c:  # (one configuration of the sequence, for a closer look)
        continue
    hp, hn, hex_, hocc, hdims, horig = frontend.plan_batch(cloud, cells, res, center, 0.0, height, infl, starts, goals, max_points=max_poly + 1,
                                                          max_vertex_dist=mvd, max_poly=max_poly, want_grid=True)
    vmap.read(cloud, cells, res, center, 0.0, height, infl)
    ok = np.array_equal(vmap.occupancy(), hocc)
    dp, dn, dex = vmap.plan_batch(starts, goals, max_points=max_poly + 1, max_vertex_dist=mvd, max_poly=max_poly)
    if os.environ.get("ONLY_CFG"):
        print("   negative n_points of the device:", np.unique(dn[dn < 0], return_counts=True))
    lim = dn <= -2  # the device's documented limits (open entries, hashed records): reported, and left out of the comparison
    n_limit += int(lim.sum())
    ok = ok and np.array_equal(hn[~lim], dn[~lim]) and np.array_equal(hex_[~lim], dex[~lim])
    if not ok and os.environ.get("ONLY_CFG"):
        d = np.nonzero(((hn != dn) | (hex_ != dex)) & ~lim)[0]
        print("   differ:", len(d), "queries; first", [(int(i), int(hn[i]), int(dn[i]), int(hex_[i]), int(dex[i])) for i in d[:12]], "(index, host n, device n, host pops, device pops)")
        print("   device n_points values among them:", np.unique(dn[d], return_counts=True), "max host pops", int(hex_.max()), "max host pops among differing", int(hex_[d].max()), "min", int(hex_[d].min()))
    if ok:
        for i in np.nonzero((hn > 0) & ~lim)[0]:
            if not np.array_equal(hp[i, :hn[i]], dp[i, :hn[i]]):
                ok = False
                break
    # corridors: the decomposition of the same vertices on the device against the host's
    fpp = abi.FH_MAX_FACES
    hf = np.zeros((nq, fpp, 4)); hoff = np.zeros((nq, 9), dtype=np.int32); hnp = np.zeros(nq, dtype=np.int32); hgoal = np.zeros((nq, 3))
    frontend.lib().ff_corridor_batch(abi.ptr(frontend._c(cloud)), len(cloud), cells[0], cells[1], cells[2], res, abi.ptr(frontend._c(center)), 0.0, height,
                                     infl, 0.05, abi.ptr(frontend._c(starts)), abi.ptr(frontend._c(goals)), nq, max_poly, mvd, fpp, abi.ptr(hf),
                                     abi.ptr(hoff), abi.ptr(hnp), abi.ptr(hgoal))
    df, doff, dnp, dgoal, tinfo = frontend.corridor_batch_device(ctx, vmap, cloud, cells, res, center, height, infl, starts, goals, max_poly, mvd, fpp, 0.05,
                                                                 search=search)
    # (the corridors come from the asynchronous device-pointer planner, which REPORTS a query that overflows the default hashed records
    # (-2) where the host-pointer entry point above runs it again with per-cell records: left out of the corridor comparison, and counted)
    lim_dev = frontend.corridor_batch_device.last_n_points <= -2
    n_limit_dev += int((lim_dev & ~lim).sum())
    lim = lim | lim_dev
    same_np = np.array_equal(hnp[~lim], dnp[~lim])
    rows_ok = same_np
    if same_np:
        sel = (hnp > 0) & ~lim
        rows_ok = np.array_equal(hoff[sel], doff[sel])
        if rows_ok:
            for i in np.nonzero(sel)[0]:
                k = hoff[i, hnp[i]]
                if not np.array_equal(hf[i, :k], df[i, :k]):
                    rows_ok = False
                    break
    bad += 0 if (ok and rows_ok) else 1
    tot_q += nq
    tot_exp += int(hex_.sum())
    print("cfg %2d side %4.0f res %.2f infl %.2f dens %.2f mvd %.1f P<=%d | grid %s paths %.2f | search %s corridors %s%s | %ds" % (
        c, side, res, infl, dens, mvd, max_poly, tuple(int(v) for v in hdims), (hn > 0).mean(), "OK" if ok else "MISMATCH",
        "OK" if rows_ok else "MISMATCH", (" | at a limit: %s (host: n_points %s, pops %s)" % (dn[lim].tolist(), hn[lim].tolist(), hex_[lim].tolist())) if lim.any() and lim.sum() < 8 else "",
        time.time() - t0), flush=True)
print("PATH SWEEP DONE (%s, record slots %d): %d configurations, %d queries, %d expanded cells, %d queries at a limit of the device search (n_points -2, not compared), %d more at the limit of the hashed records in the device-pointer planner only (run again with per-cell records by the host-pointer entry point: paths compared, corridors not), %d configurations with a mismatch" % (search, slots, ncfg, tot_q, tot_exp, n_limit, n_limit_dev, bad))
