"""Histogram of the largest number of active rows per problem (FH_QMAX_STAT build: reported in fh_result.kflops) on the C4 batch, the
C5 forest batch and the replan chain's problems:  FASTERHIP_SO=build/libfasterhip_qmax.so python scripts/r6/qmax_stat.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_amd import abi, capi, corridor, frontend, build as fb

torch.cuda.init()
dev = torch.device("cuda:0")
par = abi.default_params()


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


def pairs(whole, faces, N, shrink, max_safe_poly, rule=None, margin=None):
    B = len(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    ctx = capi.Context(0)
    ctx.set_params(par)
    if margin is not None:
        ctx.set_pair_margin(margin)
    if rule:
        ctx.set_pair_rule(**rule)
    tmpl = corridor.safe_templates(whole)
    d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(tmpl)
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros_like(d_wr)
    ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, shrink, max_safe_poly, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
    ctx.sync()
    w, s = d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype)
    ctx.close()
    return w["kflops"].copy(), s["kflops"].copy()


def show(name, q, n):
    h = np.bincount(q, minlength=n + 1)
    tail = {int(k): int(h[k:].sum()) for k in range(max(n - 16, 0), len(h)) if h[k:].sum()}
    print(json.dumps({"what": name, "problems": int(len(q)), "max": int(q.max()), "mean": float(q.mean()), "problems_with_qmax_at_least": tail}))


whole, faces, _ = corridor.whole_batch(32768, seed=3, n_seg=10, p_choices=tuple(range(2, 7)))
w, s = pairs(whole, faces, 10, 0.2, 3, margin=0.05)
show("C4 whole (n = 21)", w, 21); show("C4 safe (n = 24)", s, 24)
for seed in (11, 12):
    whole, faces, _ = corridor.whole_batch(32768, seed=seed, n_seg=10, p_choices=tuple(range(2, 7)))
    w, s = pairs(whole, faces, 10, 0.2, 3, margin=0.05)
    show("C4-like seed %d whole" % seed, w, 21); show("C4-like seed %d safe" % seed, s, 24)
fb.build_frontend()
fctx, fmap = capi.Context(0), capi.Map(0)
whole, faces, finfo = frontend.forest_batch(65536, seed=5, n_seg=15, max_poly=8, front="device", ctx=fctx, vmap=fmap, device=0, search="jps")
fmap.close(); fctx.close()
w, s = pairs(whole, faces, 15, 0.0, 5, margin=0.05, rule=dict(mode=1, r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5))
show("C5 whole (n = 36)", w, 36); show("C5 safe (n = 39)", s, 39)
w, s = pairs(whole, faces, 15, 0.2, 3, margin=0.05)
show("C5 c4-rule whole", w, 36); show("C5 c4-rule safe", s, 39)
