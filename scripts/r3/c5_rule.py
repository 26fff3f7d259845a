"""C5 hand-off variants: R rule x number of safe polytopes (solo launches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_amd import abi, capi, corridor, frontend
dev = torch.device("cuda", 0)
ctx, vmap = capi.Context(0), capi.Map(0)
frontend.forest_batch(256, seed=5, n_seg=15, max_poly=8, front="device", ctx=ctx, vmap=vmap)
whole, faces, info = frontend.forest_batch(65536, seed=5, n_seg=15, max_poly=8, front="device", ctx=ctx, vmap=vmap)
vmap.close()
B = len(whole); tmpl = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(tmpl)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev); d_sr = torch.zeros_like(d_wr)
ctx.set_pair_margin(0.05)
par = abi.default_params(); par["deadline_ms"] = 3000.0; ctx.set_params(par)
for mode, r_known, msp, shrink in ((0, 4.0, 3, 0.0), (1, 4.0, 3, 0.0), (1, 4.0, 3, 0.05), (1, 3.0, 3, 0.0), (1, 4.0, 4, 0.0), (1, 4.0, 5, 0.0)):
    ctx.set_pair_rule(mode=mode, r_known=r_known, drone_radius=0.3)
    d_safe.copy_(to_dev(tmpl))
    ms = []
    for k in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, 15, mf, 0.5, shrink, msp, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync(); ms.append(1e3 * (time.perf_counter() - t))
    w, s = d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype)
    sp = d_safe.cpu().numpy().view(abi.problem_dtype)
    live = (sp["n_seg"] > 0).sum()
    print("shrink %.2f mode %d r_known %.1f max_safe_poly %d: %.1f ms (%.0f pairs/s) safe problems %d solved %.3f iters/pair %.1f (whole %.1f safe %.1f) mean safe P %.2f trials safe %.2f" % (
        shrink, mode, r_known, msp, np.median(ms[1:]), B / (np.median(ms[1:]) * 1e-3), live, s["solved"].sum() / max(live, 1), w["qp_iters"].mean() + s["qp_iters"].mean(),
        w["qp_iters"].mean(), s["qp_iters"].mean(), sp["n_poly"][sp["n_seg"] > 0].mean(), s["trials"][sp["n_seg"] > 0].mean()), flush=True)
