import numpy as np, torch
from faster_amd import capi, frontend
n = 2048
cloud, cells, center, starts, goals = frontend.forest_queries(n, 5)
m = capi.Map(0); m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3)
dp, dn, ex = m.plan_batch(starts, goals, max_points=64)
i = int(np.argmax(ex))
t = dp[i].reshape(-1)[12:19]
print("heaviest query", i, "expansions", ex[i], "cycles per expansion by phase [reduce, settle, loads+scan, newmin+remove, relax, insert]:", np.round(t[:6] / t[6]), "total", t[:6].sum() / t[6])
j = np.argsort(ex)[n // 2]
t = dp[j].reshape(-1)[12:19]
print("median query", j, "expansions", ex[j], np.round(t[:6] / max(t[6], 1)))
