"""Per-phase cycle breakdown of fh::solve_kernel (diagnostic build with -DFH_PROFILE; not part of the product).
   FASTERHIP_EXTRA_FLAGS=-DFH_PROFILE python -c "from faster_amd import build; build.build_device(force=True)"   (or an -o elsewhere + FASTERHIP_SO)
   python scripts/phase_profile.py [pairs] [plain|pairs] [workgroups_per_cu: 0 the library's choice, 8 the two-wavefront build]
The C4 batch of bench.py (seed 3) through the fused pair kernel (default) or the whole problems alone; every problem record of the
launch carries its own cycle counters in the unused coefficient rows 12..15 (N <= 12)."""
import os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import abi, capi, corridor

names = ["staging (record, faces, LDS init)", "setup_trial", "states+CP", "scan", "build_g", "project", "backsolve+ratio+update", "add_row",
         "drop_row", "analyze", "snap save", "snap restore", "(whole problem)", "screen_constant_rows", "dt_initial", "(search() in total)",
         "look-around + donations", "result write", "hand-off of the pair (glue)", "ticket + order fetch", "child order + bounds", "leaf bookkeeping",
         "  of ticket: take_task (a frame pending?)", "  of the hand-off: drain of its stores"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
mode = sys.argv[2] if len(sys.argv) > 2 else "pairs"
ctx = capi.Context(0, pair_outputs=bool(int(os.environ.get("FH_PAIR_OUTPUTS", "0"))))  # (the library default: lazy pair outputs; FH_PROFILE builds write all 16 rows)
if len(sys.argv) > 3 and int(sys.argv[3]):
    ctx.set_sched(workgroups_per_cu=int(sys.argv[3]))
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))


def report(tag, res, min_nodes=0):
    res = res[(res["trials"] > 0) & (res["nodes"] >= min_nodes)]
    if len(res) == 0:
        return 0.0
    M = abi.FH_MAX_SEG
    prof = np.concatenate([res["coeff"][:, M - 1, :12], res["coeff"][:, M - 3, :12]], axis=1)
    cnt = np.concatenate([res["coeff"][:, M - 2, :12], res["coeff"][:, M - 4, :12]], axis=1).sum(axis=0)
    whole_problem = prof[:, 12].copy()
    search_total = prof[:, 15].copy()
    prof[:, 12] = 0
    prof[:, 15] = 0
    tot = prof.sum(axis=0)
    n = len(res)
    print("%s: %d problems; cycles per problem (mean): run_problem %.0f (+ ticket %.0f + glue %.0f outside it); iters %.1f nodes %.1f trials %.2f solved %.3f"
          % (tag, n, whole_problem.mean(), prof[:, 19].mean(), prof[:, 18].mean(), res["qp_iters"].mean(), res["nodes"].mean(), res["trials"].mean(), res["solved"].mean()))
    denom = whole_problem.sum() + prof[:, 18].sum() + prof[:, 19].sum()
    for k, (nm, v, c) in enumerate(zip(names, tot, cnt)):
        if nm.startswith("(") or nm == "-":
            continue
        print("   %-36s %6.1f%%  %9.0f cyc/problem  %7.2f calls/problem  %7.0f cyc/call" % (nm, 100 * v / denom, v / n, c / n, v / max(c, 1)))
    gp = res["coeff"][:, M - 5, :4].mean(axis=0)
    if gp.sum() > 0:
        print("   hand-off parts (cycles per pair): up to the clock %.0f | the clock loop %.0f | R, face load + polytope test %.0f | face copy %.0f" % tuple(gp))
    inner = prof[:, 2:12].sum(axis=1) + prof[:, 13] + prof[:, 16] + prof[:, 20] + prof[:, 21]
    outside = whole_problem - search_total - prof[:, 0] - prof[:, 1] - prof[:, 14] - prof[:, 17]
    print("   search() %.0f cyc/problem, of which outside its slots (stack bookkeeping, incumbent polls) %.0f; run_problem outside search / staging / dt / set-up / result write %.0f"
          % (search_total.mean(), (search_total - inner).mean(), outside.mean()))
    p2 = res["coeff"][:, M - 6, :12].mean(axis=0)
    if p2.sum() > 0:  # [r6] the finer slots (cycles per problem)
        slot = lambda k: prof[:, k].mean()
        print("   [r6] search(): prologue %.0f (of it screening %.0f) | backtrack blocks %.0f (of it restores %.0f) | before a node: limits, look, incumbent poll %.0f "
              "(of it look-around %.0f) | qp_run %.0f (of it its slots 2-8: %.0f -> bind_assignment, cost + certificates outside scan, flop count %.0f)"
              % (p2[0], slot(13), p2[1], slot(11), p2[2], slot(16), p2[3], prof[:, 2:9].sum(axis=1).mean(), p2[3] - prof[:, 2:9].sum(axis=1).mean()))
        print("   [r6] staging: record + bad_input + LDS init %.0f | basis %.0f | faces %.0f ; trial loop %.0f (set-up %.0f + search %.0f + rest %.0f) ; "
              "between the loop and the result write %.0f" % (p2[4], p2[5], slot(0) - p2[4] - p2[5], p2[6], slot(1), search_total.mean(),
              p2[6] - slot(1) - search_total.mean(), p2[7]))
        print("   [r6] ticket phase: up to take_task %.0f | pool / draw %.0f | order word %.0f" % (p2[8], p2[9], p2[10]))
        for k, nm in ((8, "up to take_task"), (9, "pool / draw"), (10, "order word")):
            v = res["coeff"][:, M - 6, k]
            if v.sum() > 0:
                print("        %-16s percentiles 10/50/90/99/max: %s ; share of the slot's cycles in its top 1 %% of problems: %.2f" % (
                    nm, " ".join("%.0f" % x for x in np.percentile(v, [10, 50, 90, 99, 100])), np.sort(v)[-max(len(v) // 100, 1):].sum() / v.sum()))
        for k, nm in ((0, "staging"), (17, "result write"), (18, "hand-off"), (19, "ticket phase")):
            v = prof[:, k]
            if v.sum() > 0:
                print("        %-16s percentiles 10/50/90/99/max: %s" % (nm, " ".join("%.0f" % x for x in np.percentile(v, [10, 50, 90, 99, 100]))))
    return denom / n


if mode == "plain":
    report("whole problems (plain launch)", ctx.solve_batch(whole, faces))
else:
    import torch
    dev = "cuda:0"
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)
    safe_t = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(safe_t)
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros_like(d_wr)
    ctx.set_pair_margin(0.05)
    for _ in range(2):
        ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, 10, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        ctx.sync()
    ms = ctx.timing_read()
    a = report("whole problems of the pairs", d_wr.cpu().numpy().view(abi.result_dtype).copy())
    b = report("safe problems of the pairs", d_sr.cpu().numpy().view(abi.result_dtype).copy())
    report("the HARD whole problems (>= 40 nodes)", d_wr.cpu().numpy().view(abi.result_dtype).copy(), min_nodes=40)
    print("per pair: %.0f cycles in the slots; launches %s ms" % (a + b, [round(float(x), 3) for x in ms[-2:]]))
