"""Per-problem work of the C5 pairs (whole and safe: solved, trials, nodes, active-set iterations) -> gpurun_out/c5_dist.npz"""
import os, sys
import numpy as np
import torch
from faster_amd import abi, capi, corridor, frontend




pairs, N = 65536, 15
dev = torch.device("cuda:0")
fctx, fmap = capi.Context(0), capi.Map(0)
whole, faces, finfo = frontend.forest_batch(pairs, seed=5, n_seg=N, max_poly=8, front="device", ctx=fctx, vmap=fmap, device=0, search="jps")
fmap.close()
B = len(whole)
tmpl = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(tmpl)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
d_sr = torch.zeros_like(d_wr)
par = abi.default_params() if hasattr(abi, "default_params") else None
if par is not None:
    fctx.set_params(par)
fctx.set_pair_margin(float(os.environ.get("R_MARGIN", "0.0")))
fctx.set_pair_rule(mode=1, r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5)
fctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.0, 5, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
fctx.sync()
w, s = d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype)
sp = d_safe.cpu().numpy().view(abi.problem_dtype)
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/c5_dist.npz", w_solved=w["solved"], w_trials=w["trials"], w_nodes=w["nodes"], w_it=w["qp_iters"], w_status=w["status"],
         s_solved=s["solved"], s_trials=s["trials"], s_nodes=s["nodes"], s_it=s["qp_iters"], s_status=s["status"], s_nseg=sp["n_seg"], s_npoly=sp["n_poly"],
         w_npoly=whole["n_poly"], mf=mf)
print("done", B, mf, w["qp_iters"].mean(), s["qp_iters"].mean())
