"""Path search: several batches in flight (one capi.Map each, own stream and workspace) — does the tail of one batch overlap with the
bulk of the next?  usage: path_overlap.py [n_per_batch] [n_maps]"""
import sys
import time

import numpy as np
import torch

from faster_amd import capi, frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
maps, bufs = [], []
for i in range(k):
    cloud, cells, center, starts, goals = frontend.forest_queries(n, 5 + i)
    m = capi.Map(0)
    m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3)
    d_s, d_g = torch.from_numpy(starts).to(dev), torch.from_numpy(goals).to(dev)
    d_p = torch.zeros((n, 9, 3), dtype=torch.float64, device=dev)
    d_n = torch.zeros(n, dtype=torch.int32, device=dev)
    d_e = torch.zeros(n, dtype=torch.int64, device=dev)
    maps.append(m)
    bufs.append((d_s, d_g, d_p, d_n, d_e))
def launch(i):
    d_s, d_g, d_p, d_n, d_e = bufs[i]
    maps[i].plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), n, 9, d_p.data_ptr(), d_n.data_ptr(), d_e.data_ptr(), 1.5, 8)
for i in range(k):
    launch(i); maps[i].sync()   # allocations
for conc in range(1, k + 1):
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 4
    for r in range(reps):
        for i in range(conc):
            launch(i)
    for i in range(conc):
        maps[i].sync()
    dt = time.perf_counter() - t
    print("%d batches in flight: %.1f ms per batch of %d, %.0f queries/s" % (conc, 1e3 * dt / (reps * conc), n, reps * conc * n / dt), flush=True)
