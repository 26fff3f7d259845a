"""Small batches through the host-pointer entry point (what one vehicle, or a few, see): median time of fh_solve_batch for n problems.
usage: small_batches.py   (GPU box; FASTERHIP_SO selects the library)"""
import sys, os, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import capi, corridor
torch.cuda.init()
c = capi.Context(0)
for n_seg, pch in ((6, (2, 3)), (10, (3, 4, 5, 6))):
    for n in (1, 16, 256, 2048):
        pr, faces, _ = corridor.make_batch(n, n_seg, pch, True, 42 + n)
        c.solve_batch(pr, faces)
        ts = []
        for rep in range(60):
            t = time.perf_counter(); r = c.solve_batch(pr, faces); ts.append(time.perf_counter() - t)
        print("N=%2d n=%5d: median %.3f ms  p90 %.3f ms  solved %.2f" % (n_seg, n, 1e3 * np.median(ts), 1e3 * np.percentile(ts, 90), (r["solved"] == 1).mean()), flush=True)
