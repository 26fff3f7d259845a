import numpy as np, torch
from faster_amd import capi, frontend
n = 65536
cloud, cells, center, starts, goals = frontend.forest_queries(n, 21)
m = capi.Map(0); m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3); m.set_search("jps")
p, k, e = m.plan_batch(starts, goals)
e = e[k > 0]
names = ["pop", "candidates", "unresolved", "cell load", "relax"]
tot = 0
for i, nm in enumerate(names):
    v = ((e >> (12 * i)) & 0xfff).mean(); tot += v
    print("%-12s %.0f ticks/pop" % (nm, v))
print("total", tot, "(s_memtime ticks: 100 MHz => %.2f us/pop)" % (tot / 100.0))
