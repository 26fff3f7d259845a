"""Device jump point search against the host restatement (plan_path_jps) on the forest of config C5, and timing of both searches."""
import sys, time
import numpy as np
import torch  # noqa: F401  (before the HIP library)
from faster_amd import capi, frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
res, infl, zmax = 0.2, 0.3, 3.0
cloud, cells, center, starts, goals = frontend.forest_queries(n, 21)
frontend.set_search("jps")
t0 = time.time()
hp, hn, hex_ = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals)
t_host = time.time() - t0
frontend.set_search("astar")
m = capi.Map(0)
m.read(cloud, cells, res, center, 0.0, zmax, infl)
m.plan_batch(starts[:64], goals[:64])
t0 = time.time(); ap, an, aex = m.plan_batch(starts, goals); t_astar = time.time() - t0
m.set_search("jps")
m.plan_batch(starts[:64], goals[:64])
t0 = time.time(); dp, dn, dex = m.plan_batch(starts, goals); t_jps = time.time() - t0
print("host jps %.3f s, device astar %.3f s, device jps %.3f s (n=%d)" % (t_host, t_astar, t_jps, n))
print("counts equal", np.array_equal(hn, dn), "expansions equal", np.array_equal(hex_, dex), "mean exp", hex_.mean(), dex.mean())
bad = 0
for i in range(n):
    if hn[i] != dn[i] or (hn[i] > 0 and not np.array_equal(hp[i, :hn[i]], dp[i, :hn[i]])):
        bad += 1
        if bad <= 5:
            print("query", i, "host", hn[i], hex_[i], "dev", dn[i], dex[i])
print("mismatching queries:", bad, "of", n, "; limit hits", int((dn == -2).sum()), "; astar found", int((an > 0).sum()), "jps found", int((dn > 0).sum()))
m.set_search("astar")
ap2, an2, aex2 = m.plan_batch(starts, goals)
print("astar after switching back identical:", np.array_equal(an, an2) and np.array_equal(ap, ap2) and np.array_equal(aex, aex2))
print("an eq", np.array_equal(an, an2), "aex eq", np.array_equal(aex, aex2), "ap eq", np.array_equal(ap, ap2))
d = np.nonzero((an != an2) | (aex != aex2))[0]
print("differing", len(d), d[:10], an[d[:10]], an2[d[:10]], aex[d[:10]], aex2[d[:10]])
for i in range(n):
    if an[i] > 0 and not np.array_equal(ap[i, :an[i]], ap2[i, :an[i]]):
        print("path differs", i, an[i]); break
ap3, an3, aex3 = m.plan_batch(starts, goals)
print("third astar equals first:", np.array_equal(an, an3) and np.array_equal(aex, aex3), "equals second:", np.array_equal(an2, an3) and np.array_equal(aex2, aex3))
