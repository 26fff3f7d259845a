"""Where the branch and bound spends its work, on the CPU restatement (no GPU needed): nodes and active-set iterations per problem of
BASELINE's C4 / C5 workloads, split into solved / unsolved problems and into the trial that succeeds / the trials that fail.
This is the harness the branching rule of DESIGN.md 4a was found with: point ORACLE_SO at a modified build of
oracle/faster_oracle.c to compare a variant (same ABI) against the committed oracle on the same problems.

    python tests/tools/bnb_stats.py [pairs=2048]
"""
import ctypes
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, corridor, frontend  # noqa: E402
from oracle import oracle, pair_glue  # noqa: E402


def solver(so):
    if not so:
        return oracle.solve_batch
    L = ctypes.CDLL(so)
    L.orc_solve_batch_mt.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_int]

    def run(pr, faces):
        pr, faces = np.ascontiguousarray(pr), np.ascontiguousarray(faces)
        res = np.zeros(len(pr), dtype=abi.result_dtype)
        par = abi.default_params()
        L.orc_solve_batch_mt(abi.ptr(pr), abi.ptr(faces), abi.ptr(par.reshape(1)), len(pr), abi.ptr(res), 0)
        return res
    return run


def report(name, solve, pr, faces, base=None):
    pr = np.ascontiguousarray(pr[pr["n_seg"] > 0])
    t = time.time()
    r = solve(pr, faces)
    dt = time.time() - t
    s = r["solved"] == 1
    p2 = pr.copy()
    p2["f_init"][s] = r["factor"][s]          # the trial that succeeds alone
    r2 = solve(p2, faces)
    u = ~s
    print("%-9s %5d problems %5.1fs | solved %.3f trials %.2f | nodes %.1f iters %.1f | solved: nodes %.1f iters %.1f (last trial alone %.1f / %.1f) | "
          "unsolved: nodes %.1f iters %.1f | share of the iterations in unsolved problems %.2f" % (
              name, len(pr), dt, s.mean(), r["trials"].mean(), r["nodes"].mean(), r["qp_iters"].mean(),
              r["nodes"][s].mean() if s.any() else 0, r["qp_iters"][s].mean() if s.any() else 0,
              r2["nodes"][s].mean() if s.any() else 0, r2["qp_iters"][s].mean() if s.any() else 0,
              r["nodes"][u].mean() if u.any() else 0, r["qp_iters"][u].mean() if u.any() else 0,
              r["qp_iters"][u].sum() / max(1, r["qp_iters"].sum())), flush=True)
    if base is not None:
        same = np.array_equal(base["solved"], r["solved"]) and np.array_equal(base["trials"], r["trials"])
        both = s & (base["solved"] == 1)
        rel = np.abs(base["cost"][both] - r["cost"][both]) / np.maximum(1e-9, np.abs(base["cost"][both]))
        print("          against the committed oracle: solved / trials identical %s, worst relative cost difference %.1e" % (same, rel.max() if both.any() else 0))
    return r


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    variant = os.environ.get("ORACLE_SO")
    sets = []
    whole, faces, _ = corridor.whole_batch(n, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    wres = oracle.solve_batch(whole, faces)
    safe, sfaces = pair_glue.glue(whole, wres, faces, corridor.safe_templates(whole), r_frac=0.5, shrink=0.2, max_safe_poly=3, r_margin=0.05)
    sets += [("C4 whole", whole, faces), ("C4 safe", safe, sfaces)]
    whole, faces, _ = frontend.forest_batch(n, seed=5, n_seg=15, max_poly=8, front="host", search="jps")
    wres = oracle.solve_batch(whole, faces)
    safe, sfaces = pair_glue.glue(whole, wres, faces, corridor.safe_templates(whole), r_frac=0.5, shrink=0.0, max_safe_poly=5, r_margin=0.05,
                                  rule=dict(r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5))
    sets += [("C5 whole", whole, faces), ("C5 safe", safe, sfaces)]
    for name, pr, fc in sets:
        base = report(name, oracle.solve_batch, pr, fc)
        if variant:
            report(name + "*", solver(variant), pr, fc, base=base)


if __name__ == "__main__":
    main()
