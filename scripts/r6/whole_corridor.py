"""The whole-corridor stage of the replan chain alone (K4 against occupied points only): python scripts/r6/whole_corridor.py [pairs]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_amd import abi, capi, frontend
torch.cuda.init()
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cloud, cells, center, starts, goals, rng = frontend.forest_queries(B, 7, return_rng=True)
def to_dev(a): return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
ctx, vmap = capi.Context(0), capi.Map(0)
mp, max_poly, fpp = 16, 3, 96
d_cloud, d_starts, d_goals = to_dev(cloud), to_dev(starts), to_dev(goals)
f64, i32 = torch.float64, torch.int32
d_paths, d_np, d_ex = torch.zeros((B, mp, 3), dtype=f64, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=torch.int64, device=dev)
d_wf = torch.zeros(B * fpp * abi.face_dtype.itemsize, dtype=torch.uint8, device=dev)
d_off, d_npoly, d_last = torch.zeros((B, 9), dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros((B, 3), dtype=f64, device=dev)
vmap.set_search("jps"); vmap.set_sphere(4.0)
vmap.read_device(d_cloud.data_ptr(), len(cloud), cells, 0.2, center, 0.0, 3.0, 0.3)
vmap.plan_batch_device(d_starts.data_ptr(), d_goals.data_ptr(), B, mp, d_paths.data_ptr(), d_np.data_ptr(), d_ex.data_ptr(), 1.5, 0)
vmap.sync()
ts = []
for _ in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), d_paths.data_ptr(), d_np.data_ptr(), B, mp, max_poly, fpp, d_wf.data_ptr(), d_off.data_ptr(),
                              d_npoly.data_ptr(), d_last.data_ptr(), 0.05, 0.0)
    ctx.sync(); ts.append(1e3 * (time.perf_counter() - t))
print(json.dumps({"whole_corridor_ms": float(np.median(ts[1:])), "checksum": int(d_wf.view(torch.int64).sum().item())}))
vmap.close(); ctx.close()
