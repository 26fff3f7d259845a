"""plan_path_jps (the host restatement of jps3d's jump point search, faster_amd/host/corridor_frontend.cpp) against the reference's OWN
compiled jps3d (oracle/_ref/libref_frontend.so: thirdparty/jps3d/src/jps_planner/{graph_search,jps_planner}.cpp untouched, driven as
JPS_Manager::solveJPS3D drives them) on randomized maps: forests at several resolutions / inflations, and cubic maps with blobs.
CPU only; needs /root/reference (the build of oracle/_ref).  usage: PYTHONPATH=. python tests/tools/jps_ref_sweep.py [queries_per_map] [maps]"""
import sys
import time

import numpy as np

from faster_amd import frontend
from oracle.ref_frontend import ref

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 150
nmaps = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rng = np.random.default_rng(77)
tot = found = bad = 0
t0 = time.time()
for k in range(nmaps):
    res = float(rng.choice([0.15, 0.2, 0.25, 0.3]))
    infl = float(rng.choice([0.0, 0.2, 0.3, 0.45, 0.6]))
    if k % 3 == 2:   # a map as high as wide with random blobs: space-diagonal jumps, goals above and below
        side = 10.0
        cells, center, zmax = (int(side / res), int(side / res), int(side / res)), np.array([side / 2] * 3), side
        blobs = rng.uniform(0.5, side - 0.5, (int(rng.integers(20, 80)), 3))
        cloud = (blobs[:, None, :] + rng.normal(0, 0.25, (len(blobs), 40, 3))).reshape(-1, 3)
        starts, goals = rng.uniform(0.3, side - 0.3, (nq, 3)), rng.uniform(0.3, side - 0.3, (nq, 3))
        kind = "blobs"
    else:
        side, height = float(rng.choice([10.0, 16.0, 22.0])), float(rng.choice([2.0, 3.0]))
        cloud, _ = frontend.forest_cloud(1000 + k, size=(side, side, height), density=float(rng.choice([0.05, 0.1, 0.2, 0.3])))
        cells, center, zmax = (int(side / res) + 6, int(side / res) + 6, int(height / res)), np.array([side / 2, side / 2, height / 2]), height
        starts = np.column_stack([rng.uniform(0.3, side - 0.3, nq), rng.uniform(0.3, side - 0.3, nq), rng.uniform(-0.2, height, nq)])
        goals = np.column_stack([rng.uniform(0.3, side - 0.3, nq), rng.uniform(0.3, side - 0.3, nq), rng.uniform(-0.2, height, nq)])
        kind = "forest"
    cloud = cloud.astype(np.float32).astype(np.float64)   # pcl::PointXYZ holds floats
    m = ref.Map(cloud, cells, res, center, 0.0, zmax, infl)
    mis = ok = 0
    for i in range(nq):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        hp, hcost, _ = frontend.plan_jps(cloud, cells, res, center, 0.0, zmax, infl, starts[i], goals[i])
        tot += 1
        if (p is None) != (hp is None) or (p is not None and (len(p) != len(hp) or not np.array_equal(p, hp))):
            mis += 1
        ok += p is not None
    m.close()
    found += ok
    bad += mis
    print("map %2d %-6s cells %-14s res %.2f inflation %.2f | %3d of %d queries have a path | vertex lists differing from the reference: %d | %ds"
          % (k, kind, tuple(cells), res, infl, ok, nq, mis, time.time() - t0), flush=True)
print("JPS REFERENCE SWEEP DONE: %d maps, %d queries, %d with a path, %d vertex lists differ from the reference's compiled jps3d" % (nmaps, tot, found, bad))
