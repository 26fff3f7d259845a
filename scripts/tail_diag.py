"""Diagnostic (FH_SHARE_PROFILE build): which pairs finish last in a fused C4 launch, and what they look like."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faster_amd import abi, capi, corridor
B, N = 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
tmpl = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = "cuda:0"
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces = to_dev(whole), to_dev(faces)
ctx = capi.Context(0); ctx.set_stream(torch.cuda.current_stream().cuda_stream); ctx.set_pair_margin(0.05)
d_safe, d_sf = to_dev(tmpl), torch.zeros_like(d_faces)
d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev); d_sr = torch.zeros_like(d_wr)
for rep in range(3):
    ctx.timing_reset()
    ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
    ctx.sync()
print("launch ms", ctx.timing_read())
w = d_wr.cpu().numpy().view(abi.result_dtype); s = d_sr.cpu().numpy().view(abi.result_dtype)
wend, wbeg, wsh, w0 = w["coeff"][:, 15, 11], w["coeff"][:, 15, 10], w["coeff"][:, 15, 9], w["coeff"][:, 15, 8]
send, sbeg, ssh, s0 = s["coeff"][:, 15, 11], s["coeff"][:, 15, 10], s["coeff"][:, 15, 9], s["coeff"][:, 15, 8]
print("pairs finished by (us): p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f" % tuple(np.percentile(send, [50, 90, 99, 99.9, 100])))
order = np.argsort(-send)[:12]
for i in order:
    print("pair %5d: whole %.0f..%.0f (sh %d, nodes %d it %d tr %d) | safe %.0f..%.0f = %.0f us (sh %d, nodes %d it %d tr %d solved %d)" % (
        i, w0[i], wend[i], wsh[i], w["nodes"][i], w["qp_iters"][i], w["trials"][i], s0[i], send[i], send[i] - s0[i], ssh[i], s["nodes"][i], s["qp_iters"][i], s["trials"][i], s["solved"][i]))
dur = send - s0
big = np.argsort(-dur)[:8]
print("longest safe problems: " + "; ".join("%.0f us from %.0f (sh %d nodes %d it %d)" % (dur[i], s0[i], ssh[i], s["nodes"][i], s["qp_iters"][i]) for i in big))
wd = wend - w0
big = np.argsort(-wd)[:8]
print("longest whole problems: " + "; ".join("%.0f us from %.0f (sh %d nodes %d it %d)" % (wd[i], w0[i], wsh[i], w["nodes"][i], w["qp_iters"][i]) for i in big))
hist, edges = np.histogram(send, bins=np.arange(0, send.max() + 500, 500))
print("pairs finishing per 0.5 ms:", hist.tolist())
late = send > np.percentile(send, 50) + 1000
print("pairs ending > p50+1ms: %d; of them safe shared %d, whole shared %d" % (late.sum(), int(ssh[late].sum()), int(wsh[late].sum())))
