"""The PCIe-inclusive leg of bench.py with different numbers of lanes: python scripts/r6/e2e_lanes.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
from faster_amd import abi, capi, corridor
torch.cuda.init()
dev = torch.device("cuda:0")
N, B = 10, 32768
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=tuple(range(2, 7)))
safe_t = corridor.safe_templates(whole)
max_faces = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())


class Pipe:
    pass


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


pp = Pipe()
pp.stream = torch.cuda.Stream(device=dev)
pp.ctx = capi.Context(0, compact_results=False)
pp.ctx.set_stream(pp.stream.cuda_stream)
pp.ctx.set_params(abi.default_params())
pp.ctx.set_pair_margin(0.05)
pp.d_safe = to_dev(safe_t)
pp.d_sfaces = torch.zeros(faces.view(np.uint8).size, dtype=torch.uint8, device=dev)
pp.d_wres = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
pp.d_sres = torch.zeros_like(pp.d_wres)
for lanes, batches in ((2, 6), (3, 9), (4, 12), (6, 18), (8, 24), (12, 36)):
    r = bench.e2e_leg(torch, dev, pp, whole, faces, safe_t, B, N, max_faces, reps=3, batches=batches, n_lanes=lanes)
    print(lanes, batches, round(r["step_ms_median"], 3), round(r["pairs_per_s"] / 1e6, 2), r["bytes_over_pcie_per_step"], r["packed_equals_full_records"], flush=True)
