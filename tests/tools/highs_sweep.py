"""Third-party verdicts on GPU output, larger than the suite's sample: HiGHS on the reference's mixed-integer constraint set (oracle/py_model.py:
milp_job) for C4-like pairs of several seeds — feasible at the factor the GPU reports, infeasible at every earlier factor, infeasible at all
ten factors for safe problems the GPU reports unsolved — and for N = 15 problems.  The helpers are those of tests/test_gpu_round6.py.
   PYTHONPATH=. python tests/tools/highs_sweep.py [pairs per seed] [seeds] [N = 15 problems]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np



def main():
    # (the HiGHS jobs run in a SPAWNED pool: its workers import this file, so nothing may run at import time)
    import torch  # (torch before the HIP library: one HIP runtime in the process)
    torch.cuda.init()
    from faster_amd import capi, corridor
    import test_gpu_round6 as T
    from test_gpu_round3 import fused_pairs

    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    n15 = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    ctx = capi.Context(0)
    t0 = time.time()
    total = 0
    for seed in range(600, 600 + seeds):
        whole, faces, _ = corridor.whole_batch(B, seed=seed, n_seg=10, p_choices=(2, 3, 4, 5, 6))
        wres, sres, safe, sfaces = fused_pairs(ctx, whole, faces, corridor.safe_templates(whole), 10, 0.05)
        base_w = np.maximum(ctx.dt_initial_batch(whole), 2 * whole["dc"])
        jobs = []
        for i in range(B):
            jobs += T._factor_jobs(("whole", i), whole[i], faces, wres[i], float(base_w[i]), 10, True)
        res = T._run_highs(jobs)
        T._judge(res, "seed %d: C4-like whole problems" % seed, B)
        total += len(res)
        live = np.nonzero(safe["n_seg"] > 0)[0]
        base_s = np.maximum(ctx.dt_initial_batch(safe), 2 * safe["dc"])
        jobs = []
        for i in live:  # every safe problem: the solved ones up to their factor, the unsolved ones at all ten factors
            jobs += T._factor_jobs(("safe", int(i)), safe[i], sfaces, sres[i], float(base_s[i]), 10, False)
        res = T._run_highs(jobs)
        T._judge(res, "seed %d: C4-like safe problems (%d unsolved: all ten factors)" % (seed, int((~sres["solved"][live].astype(bool)).sum())), len(live))
        total += len(res)
        print("  ... %d verdicts so far, %d s" % (total, time.time() - t0), flush=True)
    if n15 > 0:
        whole, faces, _ = corridor.whole_batch(n15, seed=77, n_seg=15, p_choices=(4, 5, 6, 7, 8))
        res15 = ctx.solve_batch(whole, faces)
        base = np.maximum(ctx.dt_initial_batch(whole), 2 * whole["dc"])
        jobs = []
        for i in range(n15):
            jobs += T._factor_jobs(("n15", i), whole[i], faces, res15[i], float(base[i]), 15, True)
        res = T._run_highs(jobs)
        T._judge(res, "N = 15, <= 8 polytopes whole problems", n15)
        total += len(res)
    print("HIGHS SWEEP DONE: %d verdicts, 0 disagreements | %d s" % (total, time.time() - t0))
    ctx.close()


if __name__ == "__main__":
    main()
