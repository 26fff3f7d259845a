"""Per-phase cycle breakdown of fh::solve_kernel (diagnostic build with -DFH_PROFILE; not part of the product).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFH_PROFILE -o /tmp/libfasterhip_prof.so faster_amd/csrc/fh_capi.hip
   FASTERHIP_SO=/tmp/libfasterhip_prof.so python scripts/phase_profile.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import abi, capi, corridor

names = ["staging (faces, LDS init)", "setup_trial", "states+CP", "scan", "build_g", "project", "backsolve+ratio+update", "add_row",
         "drop_row", "analyze", "snap save", "snap restore", "(whole problem)", "screen_constant_rows", "dt_initial", "search() in total"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = capi.Context(0)
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
res = ctx.solve_batch(whole, faces)
def report(tag, res):
    prof = np.concatenate([res["coeff"][:, abi.FH_MAX_SEG - 1, :12], res["coeff"][:, abi.FH_MAX_SEG - 3, :4]], axis=1)
    whole_problem = prof[:, 12].copy()
    prof[:, 12] = 0
    search_total = prof[:, 15].copy()
    prof[:, 15] = 0
    tot = prof.sum(axis=0)
    print(tag, "cycles per problem (mean): in the slots %.0f, whole problem %.0f; iters %.1f nodes %.1f trials %.2f" % (prof.sum(axis=1).mean(), whole_problem.mean(), res["qp_iters"].mean(), res["nodes"].mean(), res["trials"].mean()))
    cnt = np.concatenate([res["coeff"][:, abi.FH_MAX_SEG - 2, :12], res["coeff"][:, abi.FH_MAX_SEG - 3, 4:8]], axis=1).sum(axis=0)
    for n, v, c in zip(names, tot, cnt):
        print("   %-26s %6.1f%%  %9.0f cyc/problem  %7.1f calls/problem  %7.0f cyc/call" % (n, 100 * v / tot.sum(), v / len(res), c / len(res), v / max(c, 1)))
    inner = prof[:, 2:12].sum(axis=1) + prof[:, 13]
    print("   search() %.0f cyc/problem, of which outside the slots (bookkeeping, look-around, sharing) %.0f; run_problem outside search/setup/staging/dt (result write, epilogue) %.0f"
          % (search_total.mean(), (search_total - inner).mean(), (whole_problem - search_total - prof[:, 0] - prof[:, 1] - prof[:, 14]).mean()))
report("whole", res)
