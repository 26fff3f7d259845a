import numpy as np, torch
from faster_amd import capi, frontend
n = 2048
cloud, cells, center, starts, goals = frontend.forest_queries(n, 5)
hp, hn, _ = frontend.plan_batch(cloud, cells, 0.2, center, 0.0, 3.0, 0.3, starts, goals, max_points=64, max_vertex_dist=1.5, max_poly=0)
up, un, _ = frontend.plan_batch(cloud, cells, 0.2, center, 0.0, 3.0, 0.3, starts, goals, max_points=64)
m = capi.Map(0); m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3)
dp, dn, _ = m.plan_batch(starts, goals, max_points=64, max_vertex_dist=1.5, max_poly=0)
np.set_printoptions(precision=17)
shown = 0
tot = 0
for i in range(n):
    if hn[i] <= 0: continue
    d = np.abs(hp[i, :hn[i]] - dp[i, :hn[i]]).max(axis=1)
    if d.max() > 0:
        tot += 1
        if shown < 3:
            shown += 1
            k = int(np.argmax(d > 0))
            print("pair", i, "first differing vertex", k, "of", hn[i])
            print(" host  ", hp[i, k - 1], hp[i, k]); print(" device", dp[i, k - 1], dp[i, k])
            print(" unrefined path:", up[i, :un[i]])
print("pairs differing:", tot)
