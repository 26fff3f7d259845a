"""HBM bandwidth of the fillX kernel (K2, fh::sample_kernel): bytes written per second for the C4 whole batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from faster_amd import abi, capi, corridor

B, N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")
d_whole, d_faces = to_dev(whole), to_dev(faces)
d_res = torch.zeros(B * 1600, dtype=torch.uint8, device="cuda:0")
ctx = capi.Context(0)
stream = torch.cuda.Stream()  # (the default stream has handle 0 = "use the context's own stream")
torch.cuda.set_stream(stream)
ctx.set_stream(stream.cuda_stream)
ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_res.data_ptr())
ctx.sync()
res = d_res.cpu().numpy().view(abi.result_dtype)
cnt_ref = np.maximum(2, (N * res["dt"] / whole["dc"]).astype(np.int64)) * (res["solved"] == 1)
cap = int(cnt_ref.max())
d_states = torch.empty(B * cap * 96, dtype=torch.uint8, device="cuda:0")
d_counts = torch.zeros(B, dtype=torch.int32, device="cuda:0")
for rep in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.sample_batch_device(d_whole.data_ptr(), d_res.data_ptr(), B, cap, d_states.data_ptr(), d_counts.data_ptr())
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
counts = d_counts.cpu().numpy()
assert np.array_equal(counts, cnt_ref)
nbytes = int(counts.sum()) * 96
print("fillX: %d trajectories, %d samples (mean %.0f, capacity %d), %.1f MB written in %.3f ms => %.2f TB/s (%.0f %% of 8 TB/s)" % (
    B, counts.sum(), counts.mean(), cap, nbytes / 1e6, ms, nbytes / ms / 1e9, 100 * nbytes / ms / 1e9 / 8))
