"""The host restatement of JPS_Manager::cvxEllipsoidDecomp (faster_amd/host/corridor_frontend.cpp, what the device kernel K4 equals bit for
bit) against the reference's OWN DecompUtil headers compiled untouched (oracle/_ref/libref_frontend.so) on the corridors of forest paths:
every polytope compared as a SET of rows (1e-9); how many also come in the same order is reported.  CPU only by default (building
oracle/_ref needs /root/reference; the GPU box uses the prebuilt file).  With a third argument `device` the same legs are also decomposed
by the device kernel K4 (fh_decompose_batch) and its rows are compared with the host restatement's BIT FOR BIT and in order.
usage: PYTHONPATH=. python tests/tools/decomp_ref_sweep.py [paths_per_map] [maps] [device]"""
import sys
import time

import numpy as np

from faster_amd import frontend
from oracle.ref_frontend import ref

npaths = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nmaps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
device = len(sys.argv) > 3 and sys.argv[3] == "device"
ctx = None
if device:
    import torch  # noqa: F401  (one HIP runtime per process)
    from faster_amd import capi

    ctx = capi.Context(0)
dev_polys = dev_equal = 0
rng = np.random.default_rng(31)
key = lambda M: M[np.lexsort(np.round(M, 6).T[::-1])]
polys = same_set = same_order = worst = 0
t0 = time.time()
for k in range(nmaps):
    side, height = float(rng.choice([10.0, 16.0, 22.0])), float(rng.choice([2.0, 3.0]))
    res, infl = float(rng.choice([0.15, 0.2, 0.25])), float(rng.choice([0.2, 0.3, 0.45]))
    radius = float(rng.choice([0.0, 0.05, 0.2]))
    mvd, max_poly = float(rng.choice([0.8, 1.5, 2.5])), int(rng.choice([3, 5, 8]))
    cloud, _ = frontend.forest_cloud(2000 + k, size=(side, side, height), density=float(rng.choice([0.05, 0.1, 0.2])))
    cloud = cloud.astype(np.float32).astype(np.float64)
    cells, center = (int(side / res) + 6, int(side / res) + 6, int(height / res)), np.array([side / 2, side / 2, height / 2])
    starts = np.column_stack([rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.3, height - 0.3, npaths)])
    goals = np.column_stack([rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.3, height - 0.3, npaths)])
    frontend.set_search("jps")
    paths, npts, _ = frontend.plan_batch(cloud, cells, res, center, 0.0, height, infl, starts, goals, max_points=max_poly + 1, max_vertex_dist=mvd,
                                         max_poly=max_poly)
    frontend.set_search("astar")
    mp = ms = mo = 0
    for i in range(npaths):
        if npts[i] < 2:
            continue
        path = paths[i, :npts[i]]
        want = ref.decompose(path, cloud, radius, 0.0)
        got, _ = frontend.decompose(path, cloud, drone_radius=radius, z_ground=0.0)
        assert len(got) == len(want)
        if device:
            dfaces, dcounts = ctx.decompose_batch(cloud, np.hstack([path[:-1], path[1:]]), drone_radius=radius, z_ground=0.0, max_faces=96)
            for j, (A, b) in enumerate(got):
                dev_polys += 1
                rows = np.column_stack([dfaces["a"][j, :max(dcounts[j], 0)], dfaces["b"][j, :max(dcounts[j], 0)]])
                dev_equal += int(dcounts[j] == len(b) and np.array_equal(rows, np.column_stack([A, b])))
        for (A, b), (A2, b2) in zip(got, want):
            mp += 1
            if len(b) != len(b2):
                continue
            G, W = key(np.column_stack([A, b])), key(np.column_stack([A2, b2]))
            d = float(np.abs(G - W).max()) if len(b) else 0.0
            worst = max(worst, d if d < 1e-6 else 0.0)
            if d <= 1e-9:
                ms += 1
                mo += int(np.allclose(np.column_stack([A, b]), np.column_stack([A2, b2]), rtol=0, atol=1e-9))
    polys += mp; same_set += ms; same_order += mo
    print("map %2d side %2.0f res %.2f inflation %.2f radius %.2f legs<=%d of <=%.1f m | polytopes %4d | equal as sets %4d | also in the same order %4d | %ds"
          % (k, side, res, infl, radius, max_poly, mvd, mp, ms, mo, time.time() - t0), flush=True)
print("DECOMPOSITION REFERENCE SWEEP DONE: %d maps, %d polytopes, %d equal to the reference's DecompUtil as sets of rows (1e-9; worst difference %.1e), "
      "%d of them with the rows in the same order" % (nmaps, polys, same_set, worst, same_order))
if device:
    print("DEVICE K4 (fh_decompose_batch) on the same %d legs: %d equal to the host restatement bit for bit and in order" % (dev_polys, dev_equal))
    assert dev_polys == dev_equal == polys
