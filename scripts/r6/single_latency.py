"""Latency of ONE genNewTraj-sized solve through fh_solve_batch (host pointers): python scripts/r6/single_latency.py  (FASTERHIP_SO selects the library)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.cuda.init()
from faster_amd import capi, corridor
whole, faces, _ = corridor.whole_batch(256, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
ctx = capi.Context(0)
ts = []
for i in range(220):
    p = whole[i % 256:i % 256 + 1].copy()
    f0, n = int(p["face_begin"][0]), int(p["face_off"][0][p["n_poly"][0]])
    f = faces[f0:f0 + n].copy(); p["face_begin"] = 0
    t = time.perf_counter(); r = ctx.solve_batch(p, f); ts.append(1e3 * (time.perf_counter() - t))
ts = np.array(ts[20:])
print("single solve: median %.4f ms p95 %.4f ms" % (np.median(ts), np.percentile(ts, 95)))
