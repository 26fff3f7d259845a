"""One C4 launch alone on the GPU under different scheduling settings (fh_sched): what does its tail want?"""
import os, sys, time
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor

B, N = 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
safe_t = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = "cuda:0"
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(safe_t)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
d_sr = torch.zeros_like(d_wr)
ctx = capi.Context(0)
ctx.set_pair_margin(0.05)
def run(**kw):
    ctx.set_sched(**kw)
    ms = []
    for _ in range(6):
        torch.cuda.synchronize()
        ctx.timing_reset()
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync()
        ms.append(float(ctx.timing_read()[-1]))
    st = ctx.share_stats()
    print("%-60s launch ms: median %.3f min %.3f | donated %d" % (kw, np.median(ms[1:]), min(ms[1:]), st["donated"]), flush=True)
run()
for w in (64, 256, 1024, 2816):
    run(waiting_workgroups=w)
for w in (64, 256, 1024):
    run(waiting_workgroups=w, backlog=128, publish_factor=2)
run(waiting_workgroups=256, min_nodes=1)
for cu in (8, 9, 10):
    run(workgroups_per_cu=cu)
    run(workgroups_per_cu=cu, waiting_workgroups=256)
run(launch_order=0)
run(child_bound=0)
