"""When do the units of ONE C4 launch end?  (-DFH_SHARE_PROFILE build: every result carries the time its problem began / ended on its
workgroup, in us since that workgroup started.)"""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor
B, N = 32768, 10
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
safe_t = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = "cuda:0"
to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(safe_t)
d_sf = torch.zeros_like(d_faces)
d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
d_sr = torch.zeros_like(d_wr)
ctx = capi.Context(0)
ctx.set_pair_margin(0.05)
for kw in ({}, {"workgroups_per_cu": 8}):
    ctx.set_sched(**kw)
    for _ in range(3):
        ctx.timing_reset()
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync()
    ms = float(ctx.timing_read()[-1])
    w = d_wr.cpu().numpy().view(abi.result_dtype); s = d_sr.cpu().numpy().view(abi.result_dtype)
    M = abi.FH_MAX_SEG
    w_end, s_end, s_beg, w_beg = w["coeff"][:, M - 1, 11], s["coeff"][:, M - 1, 11], s["coeff"][:, M - 1, 10], w["coeff"][:, M - 1, 10]
    end = np.maximum(w_end, s_end)
    print(kw, "launch %.3f ms; pair end times (us since its workgroup started): " % ms + " ".join("p%g=%.0f" % (q, np.percentile(end, q)) for q in (50, 90, 99, 99.9, 100)))
    late = np.argsort(end)[-12:]
    print("   the 12 last pairs: end", np.round(end[late]).tolist(), "| whole dur", np.round(w_end[late] - w_beg[late]).tolist(), "| safe dur", np.round(s_end[late] - s_beg[late]).tolist(),
          "| safe iters", s["qp_iters"][late].tolist(), "trials", s["trials"][late].tolist(), "shared", s["coeff"][late, M - 1, 9].tolist())
    dur = (w_end - w_beg) + (s_end - s_beg)
    print("   pair duration us: mean %.0f p50 %.0f p99 %.0f max %.0f; sum of durations / (waves x launch) = %.2f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 99), dur.max(), dur.sum() / (ctx.share_stats()["workgroups"] * ms * 1e3)))
    longest = np.argsort(dur)[-16:]
    print("   the 16 longest pairs: dur", np.round(dur[longest]).tolist(), "| begin", np.round(w_beg[longest]).tolist(), "| whole dur", np.round((w_end - w_beg)[longest]).tolist(),
          "whole iters", w["qp_iters"][longest].tolist(), "nodes", w["nodes"][longest].tolist(), "trials", w["trials"][longest].tolist(),
          "| safe iters", s["qp_iters"][longest].tolist(), "nodes", s["nodes"][longest].tolist(), "trials", s["trials"][longest].tolist(), "solved", s["solved"][longest].tolist(),
          "shared", s["coeff"][longest, M - 1, 9].tolist(), "n_poly", whole["n_poly"][longest].tolist())
    big = dur > 500
    print("   pairs longer than 500 us: %d, their share of the work %.2f; of them safe unsolved %.2f, whole iters mean %.0f, safe iters mean %.0f" % (
        big.sum(), dur[big].sum() / dur.sum(), (s["solved"][big] == 0).mean(), w["qp_iters"][big].mean(), s["qp_iters"][big].mean()))
    # cycles per iteration-ish: duration vs iterations
    it = w["qp_iters"] + s["qp_iters"]
    print("   us per pair = %.1f + %.2f * iterations (least squares)" % tuple(np.polyfit(it, dur, 1)[::-1]))
