"""GPU diagnostic for the work-sharing branch and bound (not part of the product): one C4 batch, launched alone, with and without
sharing — launch times from HIP events, donation statistics, and bit-for-bit comparison of the results of the two modes (and of the
fused pair kernel)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from faster_amd import abi, capi, corridor

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10
PMAX = int(sys.argv[3]) if len(sys.argv) > 3 else 6
REPS = 4
MARGIN = float(os.environ.get("PAIR_MARGIN", "-1"))
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=tuple(range(2, PMAX + 1)))
safe_t = corridor.safe_templates(whole)
mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
dev = "cuda:0"


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


d_whole, d_faces = to_dev(whole), to_dev(faces)
RES = abi.result_dtype.itemsize
FIELDS = ("solved", "trials", "status", "factor", "dt", "cost", "coeff", "assign")


def same(a, b, what):
    bad = 0
    for f in FIELDS:
        ne = a[f] != b[f]
        if ne.any():
            idx = np.nonzero(ne.reshape(len(a), -1).any(axis=1))[0]
            bad += len(idx)
            print("   MISMATCH %s field %s: %d problems, first %s" % (what, f, len(idx), idx[:5]))
    return bad


out = {}
for mode in ("share0", "share1"):
    ctx = capi.Context(0)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    par = abi.default_params()
    par["share"] = 0 if mode == "share0" else 1
    ctx.set_params(par)
    ctx.set_pair_margin(MARGIN)
    d_safe = to_dev(safe_t)
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * RES, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros_like(d_wr)
    tw, ts, stw, sts = [], [], None, None
    for rep in range(REPS):
        ctx.timing_reset()
        ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_wr.data_ptr())
        stw = ctx.share_stats()
        ctx.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
        ctx.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, mf, d_sr.data_ptr())
        sts = ctx.share_stats()
        ms = ctx.timing_read()
        tw.append(ms[0]); ts.append(ms[1])
    wr = d_wr.cpu().numpy().view(abi.result_dtype).copy()
    sr = d_sr.cpu().numpy().view(abi.result_dtype).copy()
    out[mode] = (wr, sr)
    print("%s: whole launch ms %s | safe launch ms %s" % (mode, np.round(tw, 3), np.round(ts, 3)))
    print("   whole stats %s" % stw)
    print("   safe  stats %s" % sts)
    pf = ctx.share_profile().astype(np.float64)
    if pf.any():  # -DFH_SHARE_PROFILE build: 100 MHz ticks -> us (profile of the last launch = safe)
        us = lambda t, c: (t / 100.0 / max(c, 1.0), int(c))
        print("   safe profile (mean us, count): look %.1f x%d | donate %.1f x%d | wait %.1f x%d | frame copy %.1f x%d" % (
            us(pf[0], pf[1]) + us(pf[2], pf[3]) + us(pf[4], pf[5]) + us(pf[6], pf[7])))
        print("      frame set-up %.1f x%d | frame search %.1f us per frame, %.1f nodes per frame (%d nodes) | finish_part %.1f x%d | "
              "dry tail per worker %.1f us x%d" % (us(pf[8], pf[9]) + (pf[10] / 100.0 / max(pf[9], 1), pf[11] / max(pf[9], 1), int(pf[11])) +
                                                   us(pf[12], pf[13]) + us(pf[14], pf[15])))
    for name, r in (("whole", wr), ("safe", sr)):
        it = r["qp_iters"].astype(np.float64)
        top = np.argsort(-it)[:6]
        print("   %s hardest: %s" % (name, ", ".join("it %d nodes %d trials %d solved %d" % (r["qp_iters"][i], r["nodes"][i], r["trials"][i], r["solved"][i]) for i in top)))
        print("   %s: solved %.4f iters mean %.1f p99 %.0f p99.9 %.0f max %.0f | nodes mean %.2f max %d | trials mean %.2f | kflops mean %.1f" % (
            name, r["solved"].mean(), it.mean(), np.percentile(it, 99), np.percentile(it, 99.9), it.max(), r["nodes"].mean(), r["nodes"].max(),
            r["trials"].mean(), r["kflops"].mean()))
    # fused pair kernel
    d_wr2 = torch.zeros_like(d_wr); d_sr2 = torch.zeros_like(d_wr)
    d_safe2 = to_dev(safe_t); d_sf2 = torch.zeros_like(d_faces)
    tf = []
    for rep in range(REPS):
        ctx.timing_reset()
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr2.data_ptr(), d_safe2.data_ptr(),
                               d_sf2.data_ptr(), d_sr2.data_ptr())
        stf = ctx.share_stats()
        tf.append(ctx.timing_read()[0])
    print("   fused pairs launch ms %s stats %s" % (np.round(tf, 3), stf))
    wr2 = d_wr2.cpu().numpy().view(abi.result_dtype).copy()
    sr2 = d_sr2.cpu().numpy().view(abi.result_dtype).copy()
    bad = same(wr, wr2, mode + " fused-vs-split whole") + same(sr, sr2, mode + " fused-vs-split safe")
    print("   fused == split: %s" % ("yes" if bad == 0 else "NO (%d)" % bad))
    ctx.close()
bad = same(out["share0"][0], out["share1"][0], "share whole") + same(out["share0"][1], out["share1"][1], "share safe")
print("share1 == share0 (solved, trials, status, factor, dt, cost, coeff, assign): %s" % ("yes" if bad == 0 else "NO (%d)" % bad))
