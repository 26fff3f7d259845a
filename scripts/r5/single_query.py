"""Latency of ONE path query through the host-pointer entry point (what JpsHip::solveJPS3D pays per replan of one vehicle):
the first query after a map update builds the jump tables of the map, the following ones reuse them."""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import capi, frontend

cloud, cells, center, starts, goals = frontend.forest_queries(64, 7)
for mode in ("jps", "astar"):
    m = capi.Map(0)
    m.set_search(mode)
    t_first, t_next, t_read = [], [], []
    for rep in range(12):
        t = time.perf_counter(); m.read(cloud, cells, 0.2, center, 0.0, 3.0, 0.3); t_read.append(1e3 * (time.perf_counter() - t))
        t = time.perf_counter(); p, n, e = m.plan_batch(starts[rep:rep + 1], goals[rep:rep + 1]); t_first.append(1e3 * (time.perf_counter() - t))
        for k in range(4):
            t = time.perf_counter(); p, n, e2 = m.plan_batch(starts[rep + 8 + k:rep + 9 + k], goals[rep + 8 + k:rep + 9 + k]); t_next.append((1e3 * (time.perf_counter() - t), int(e2[0])))
    nx = np.array([a for a, _ in t_next[4:]]); pops = np.array([b for _, b in t_next[4:]])
    print("%s: map update %.2f ms | first query after it %.2f ms median | later queries %.2f ms median (%.0f pops median, %.2f us per pop)"
          % (mode, np.median(t_read[1:]), np.median(t_first[1:]), np.median(nx), np.median(pops), 1e3 * np.median(nx) / max(1, np.median(pops))))
    m.close()
