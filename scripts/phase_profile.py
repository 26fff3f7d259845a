"""Per-phase cycle breakdown of fh::solve_kernel (diagnostic build with -DFH_PROFILE; not part of the product).
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFH_PROFILE -o /tmp/libfasterhip_prof.so faster_amd/csrc/fh_capi.hip
   FASTERHIP_SO=/tmp/libfasterhip_prof.so python scripts/phase_profile.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import abi, capi, corridor

names = ["load+dt_init", "trial setup+screen+eq", "states+CP", "scan", "build_g", "project", "backsolve+ratio+update", "add_row",
         "drop_row", "analyze", "snap save", "snap restore"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = capi.Context(0)
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
res = ctx.solve_batch(whole, faces)
def report(tag, res):
    prof = res["coeff"][:, abi.FH_MAX_SEG - 1, :12]
    tot = prof.sum(axis=0)
    print(tag, "cycles per problem (mean): total %.0f; iters %.1f nodes %.1f" % (prof.sum(axis=1).mean(), res["qp_iters"].mean(), res["nodes"].mean()))
    cnt = res["coeff"][:, abi.FH_MAX_SEG - 2, :12].sum(axis=0)
    for n, v, c in zip(names, tot, cnt):
        print("   %-26s %6.1f%%  %9.0f cyc/problem  %7.1f calls/problem  %7.0f cyc/call" % (n, 100 * v / tot.sum(), v / len(res), c / len(res), v / max(c, 1)))
report("whole", res)
