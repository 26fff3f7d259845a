"""SURVEY.md 8(f) N4 measurement: is feasibility monotone in the time-allocation factor?  Every factor of the window
[1, 10] step 1 is solved as its own single-trial problem for a sample of the C4 whole and safe workloads; reports how many
problems have a feasible factor BELOW an infeasible one (where bisection would return a different factor_that_worked_
than the reference's linear scan), and the per-factor cost of a trial."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import abi, capi, corridor

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ctx = capi.Context(0)
for tag, (pr, fc, _) in {"whole (N=10, P<=6, final position forced)": corridor.whole_batch(n, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6)),
                         "safe  (N=10, P<=3, final position free)": corridor.make_batch(n, 10, (1, 2, 3), False, 5)}.items():
    factors = np.arange(1.0, 10.5, 1.0)
    sub = np.repeat(pr, len(factors))
    sub["f_init"] = sub["f_final"] = np.tile(factors, n)
    t = time.perf_counter()
    res = ctx.solve_batch(sub, fc)
    el = time.perf_counter() - t
    feas = res["solved"].reshape(n, len(factors)).astype(bool)
    first = np.where(feas.any(axis=1), feas.argmax(axis=1), len(factors))
    holes = np.array([(~feas[i, first[i]:]).any() for i in range(n)])
    print("%s: %d problems x %d factors in %.1f ms" % (tag, n, len(factors), 1e3 * el))
    print("   feasible at some factor: %d; NON-monotone (an infeasible factor above a feasible one): %d (%.2f %%)" %
          (feas.any(axis=1).sum(), holes.sum(), 100.0 * holes.mean()))
    print("   first feasible factor histogram:", np.bincount(first, minlength=len(factors) + 1).tolist())
    nodes = res["nodes"].reshape(n, len(factors)); it = res["qp_iters"].reshape(n, len(factors))
    print("   mean B&B nodes per trial by factor:", np.round(nodes.mean(axis=0), 1).tolist())
    print("   mean QP iterations per trial by factor:", np.round(it.mean(axis=0), 1).tolist())
    print("   max  QP iterations per trial by factor:", it.max(axis=0).tolist())
