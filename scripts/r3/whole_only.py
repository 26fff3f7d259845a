"""Whole-only throughput (plain kernel, no hand-off): for ablation builds whose results are incomplete on purpose."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_amd import abi, capi, corridor
B, N = 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = torch.device("cuda", 0)
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces = to_dev(whole), to_dev(faces)
pipes = []
for k in range(8):
    st = torch.cuda.Stream(device=dev); c = capi.Context(0); c.set_stream(st.cuda_stream)
    pipes.append((c, torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)))
def step(i):
    c, r = pipes[i % len(pipes)]
    c.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, r.data_ptr())
for i in range(16): step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(64): step(i)
torch.cuda.synchronize()
el = time.perf_counter() - t
res = pipes[0][1].cpu().numpy().view(abi.result_dtype)
print("%s: %.3f M whole problems/s (%.3f ms per 32768), solved %.3f iters %.1f" % (os.path.basename(capi.SO_PATH), 64 * B / el / 1e6, 1e3 * el / 64, res["solved"].mean(), res["qp_iters"].mean()))
