import numpy as np, torch
from faster_amd import capi, frontend
n = 16384
cloud, cells, center, starts, goals = frontend.forest_queries(n, 21)
m = capi.Map(0); m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3); m.set_search("jps")
p, k, e = m.plan_batch(starts, goals)
pops = e & 0xfffff; unres = (e >> 20) & 0xfffff; rounds = e >> 40
print("per query: pops %.1f, unresolved jumps %.1f, lock-step rounds %.1f; per pop: unresolved %.2f rounds %.2f" % (pops.mean(), unres.mean(), rounds.mean(), unres.sum() / pops.sum(), rounds.sum() / pops.sum()))
