"""One GPU run that prints the figures quoted in DESIGN.md sections 4, 7b, 7c and 9 (not part of the product)."""
import json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # before libfasterhip (one HIP runtime per process)
from faster_amd import abi, capi, corridor, frontend

ctx = capi.Context(0)

def kernel_ms(pr, fc, reps=5):
    ctx.solve_batch(pr, fc)
    ts = []
    for _ in range(reps):
        ctx.solve_batch(pr, fc)
        ts.append(ctx.last_kernel_ms())
    return float(np.median(ts))

ONLY = os.environ.get("ROUND_ONLY", "")
# --- BASELINE configurations C2, C3 and the C2 generator at batch 65536
pr, fc, _ = corridor.safe_batch(1024, seed=1)
ms = kernel_ms(pr, fc); print("C2  1024 safe QPs N=6 P=1        : kernel %.3f ms  => %.1f M solves/s" % (ms, 1024 / ms / 1e3))
pr, fc, _ = corridor.safe_batch(65536, seed=1)
ms = kernel_ms(pr, fc); print("C2 generator at batch 65536      : kernel %.3f ms  => %.1f M solves/s" % (ms, 65536 / ms / 1e3))
pr, fc, _ = corridor.whole_batch(4096, seed=2, n_seg=10, p_choices=(2, 3, 4))
ms = kernel_ms(pr, fc); print("C3  4096 whole MIQPs N=10 P<=4   : kernel %.3f ms  => %.2f M solves/s" % (ms, 4096 / ms / 1e3))

# --- C4: one launch at a time (straggler tail), work distribution
B, N = 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
safe_t = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(safe_t)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * 1600, dtype=torch.uint8, device="cuda:0"); d_sr = torch.zeros_like(d_wr)
c2 = capi.Context(0)
c2.set_stream(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    c2.timing_reset()
    c2.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_wr.data_ptr())
    c2.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
    c2.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, mf, d_sr.data_ptr())
    c2.sync()
    ms2 = c2.timing_read()
print("C4 one launch at a time (32768): whole %.2f ms, safe %.2f ms" % (ms2[0], ms2[1]))
for name, d in (("whole", d_wr), ("safe", d_sr)):
    r = d.cpu().numpy().view(abi.result_dtype)
    it = r["qp_iters"].astype(np.float64)
    print("   %-5s iterations: median %.0f  p99 %.0f  p99.9 %.0f  max %.0f ; nodes mean %.2f max %d ; trials mean %.2f ; solved %.3f" % (
        name, np.percentile(it, 50), np.percentile(it, 99), np.percentile(it, 99.9), it.max(), r["nodes"].mean(), r["nodes"].max(),
        r["trials"].mean(), r["solved"].mean()))

# --- N1 on the device: convex decomposition throughput on the forest scene
cloud, _ = frontend.forest_cloud(seed=7)
rng = np.random.default_rng(1)
nseg = 16384
p1 = np.column_stack([rng.uniform(1, 19, nseg), rng.uniform(1, 19, nseg), rng.uniform(0.5, 2.5, nseg)])
dirv = rng.normal(size=(nseg, 3)); dirv /= np.linalg.norm(dirv, axis=1)[:, None]
segs = np.ascontiguousarray(np.hstack([p1, p1 + dirv * rng.uniform(0.5, 1.5, (nseg, 1))]))
ctx.decompose_batch(cloud, segs[:64])
t = time.perf_counter(); fcs, cnt = ctx.decompose_batch(cloud, segs); el = time.perf_counter() - t
print("decomposition on the device: %d segments, cloud of %d points: %.2f ms host-to-host (%.0f segments/s); rows mean %.1f, overflow %d" % (
    nseg, cloud.shape[0], 1e3 * el, nseg / el, cnt[cnt > 0].mean(), (cnt < 0).sum()))
t = time.perf_counter()
for k in range(256):
    frontend.decompose(segs[k].reshape(2, 3), cloud)
el = time.perf_counter() - t
print("decomposition, host front-end (one thread): %.0f segments/s" % (256 / el))
