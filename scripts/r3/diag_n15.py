"""Round-3 diagnostic: which problems of the C5 parity batch differ from the oracle, per kernel instantiation / sharing mode."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from faster_amd import abi, capi, corridor
from oracle import oracle as orc

pr, faces, _ = corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))
ref = orc.solve_batch(pr, faces)
ctx = capi.Context(0)
def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")
def run(max_seg, share):
    par = abi.default_params(); par["share"] = share; ctx.set_params(par)
    dp, df = dev(pr), dev(faces)
    dr = torch.zeros(len(pr) * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    mf = int(pr["face_off"].max())
    ctx.solve_batch_device(dp.data_ptr(), df.data_ptr(), len(pr), max_seg, mf, dr.data_ptr())
    ctx.sync()
    return dr.cpu().numpy().view(abi.result_dtype)
for max_seg in (15, 16):
    for share in (1, 0):
        got = run(max_seg, share)
        bad = np.nonzero((got["trials"] != ref["trials"]) | (got["solved"] != ref["solved"]))[0]
        print("max_seg", max_seg, "share", share, "mismatches", len(bad))
        for i in bad[:8]:
            print("   i", i, "P", pr["n_poly"][i], "gpu solved/trials/status/nodes/iters", got["solved"][i], got["trials"][i], got["status"][i], got["nodes"][i], got["qp_iters"][i],
                  "oracle", ref["solved"][i], ref["trials"][i], ref["status"][i], ref["nodes"][i], "cost", got["cost"][i], ref["cost"][i])
        ok = (got["solved"] == 1) & (ref["solved"] == 1) & (got["trials"] == ref["trials"])
        if ok.any():
            print("   cost rel diff max", np.max(np.abs(got["cost"][ok] - ref["cost"][ok]) / np.maximum(1e-9, np.abs(ref["cost"][ok]))), "coeff", np.max(np.abs(got["coeff"][ok] - ref["coeff"][ok])))
