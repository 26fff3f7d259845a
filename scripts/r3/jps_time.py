"""Timing of the two device searches on the forest of config C5 (device-resident queries, events around the launch)."""
import sys
import numpy as np
import torch
from faster_amd import capi, frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
res, infl, zmax = 0.2, 0.3, 3.0
cloud, cells, center, starts, goals = frontend.forest_queries(n, 21)
m = capi.Map(0)
m.read(cloud, cells, res, center, 0.0, zmax, infl)
dev = torch.device("cuda:0")
d_s = torch.from_numpy(starts).to(dev); d_g = torch.from_numpy(goals).to(dev)
d_p = torch.zeros((n, 64, 3), dtype=torch.float64, device=dev); d_n = torch.zeros(n, dtype=torch.int32, device=dev); d_e = torch.zeros(n, dtype=torch.int64, device=dev)
for mode in ("astar", "jps"):
    m.set_search(mode)
    for waves in ([20, 16, 12] if len(sys.argv) > 2 else [0]):
        m.set_sched(waves, 1)
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            m.plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), n, 64, d_p.data_ptr(), d_n.data_ptr(), d_e.data_ptr())
            m.sync()
            best = min(best, time.perf_counter() - t0)
        print("%s waves/CU %d: %.1f ms, %.0f queries/s, mean pops %.1f, found %d" % (mode, waves, best * 1e3, n / best, d_e.double().mean().item(), int((d_n > 0).sum())))
