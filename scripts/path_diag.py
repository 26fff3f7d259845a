"""Device voxel path search vs the host restatement: occupancy, vertices, expansions; timings.  usage: path_diag.py [n] [seed]"""
import sys
import time

import numpy as np
import torch  # noqa: F401  (HIP runtime first)

from faster_amd import capi, frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 5
res, infl, zmax = 0.2, 0.3, 3.0
cloud, cells, center, starts, goals = frontend.forest_queries(n, seed)
t = time.time()
hp, hn, hex_, hocc, hdims, horig = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, want_grid=True)
t_host = time.time() - t
m = capi.Map(0)
m.read(cloud, cells, res, center, 0.0, zmax, infl)
dims, orig = m.dims()
print("dims", dims, hdims, "origin equal", np.array_equal(orig, horig), flush=True)
occ = m.occupancy()
print("occupancy equal:", np.array_equal(occ, hocc), "occupied frac", (occ != 0).mean(), flush=True)
t = time.time()
dp, dn, dex = m.plan_batch(starts, goals)
t_dev0 = time.time() - t
t = time.time()
dp, dn, dex = m.plan_batch(starts, goals)
t_dev = time.time() - t
print("host %.3f s (%.0f q/s), device first %.3f s, second %.3f s (%.0f q/s)" % (t_host, n / t_host, t_dev0, t_dev, n / t_dev), flush=True)
print("n_points equal:", np.array_equal(hn, dn), " expansions equal:", np.array_equal(hex_, dex), " mean expansions", hex_.mean(), "max", hex_.max())
bad = np.nonzero(hn != dn)[0]
print("differing counts:", len(bad), bad[:10], hn[bad[:10]], dn[bad[:10]])
same = hn == dn
worst = 0.0
nbad = 0
for i in np.nonzero(same & (hn > 0))[0]:
    d = np.abs(hp[i, :hn[i]] - dp[i, :hn[i]]).max()
    worst = max(worst, d)
    nbad += d != 0
print("paths with any differing vertex:", nbad, "worst abs diff", worst)
eb = np.nonzero(hex_ != dex)[0]
print("differing expansions:", len(eb), eb[:8], hex_[eb[:8]], dex[eb[:8]])
# corridor vertices (createMoreVertexes + deleteVertexes)
hp2, hn2, _ = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8)
dp2, dn2, _ = m.plan_batch(starts, goals, max_points=16, max_vertex_dist=1.5, max_poly=8)
print("refined: counts equal", np.array_equal(hn2, dn2), "vertices equal", np.array_equal(hp2[hn2 > 0], dp2[hn2 > 0]) if np.array_equal(hn2, dn2) else None)
if not np.array_equal(hn2, dn2):
    b = np.nonzero(hn2 != dn2)[0]
    print(b[:10], hn2[b[:10]], dn2[b[:10]])
else:
    worst = 0.0
    for i in np.nonzero(hn2 > 0)[0]:
        worst = max(worst, np.abs(hp2[i, :hn2[i]] - dp2[i, :hn2[i]]).max())
    print("refined worst abs diff", worst)
