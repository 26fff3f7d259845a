"""GPU diagnostic: per-launch time and per-problem work distribution of the C4 workload (not part of the product)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faster_amd import abi, capi, corridor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
safe_t = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = "cuda:0"
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(safe_t)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * 1600, dtype=torch.uint8, device=dev); d_sr = torch.zeros_like(d_wr)
ctx = capi.Context(0)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    ctx.timing_reset()
    ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_wr.data_ptr())
    ctx.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
    ctx.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, mf, d_sr.data_ptr())
    ctx.sync()
    ms = ctx.timing_read()
print("launch ms whole/safe:", ms)
for name, d, t in (("whole", d_wr, ms[0]), ("safe", d_sr, ms[1])):
    r = d.cpu().numpy().view(abi.result_dtype)
    it = r["qp_iters"].astype(np.float64)
    nd = r["nodes"]
    print(name, "iters mean %.1f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f | nodes mean %.2f p99 %.0f max %d | trials mean %.2f" % (
        it.mean(), np.percentile(it, 50), np.percentile(it, 90), np.percentile(it, 99), np.percentile(it, 99.9), it.max(),
        nd.mean(), np.percentile(nd, 99), nd.max(), r["trials"].mean()))
    slots = 256 * 5
    print("   us per iteration if balanced over %d waves: %.2f ; max-problem iters x that = %.2f ms of the %.2f ms launch" % (
        slots, t * 1e3 * slots / it.sum(), it.max() * t * slots / it.sum(), t))
