"""Counts, on the GPU path, what Gurobi's default tolerances could change relative to this repository's exact settings (SURVEY.md
8(c): "marginal cases reported separately"): re-solves samples of BASELINE configs C2..C5 with feas_tol 1e-6 (Gurobi FeasibilityTol)
and with mip_gap 1e-4 (Gurobi MIPGap) and counts flips of `solved`, `factor_that_worked_` and the assignment.  Writes
profiles/<tag>_gurobi_semantics.json.  Run on the GPU box: python scripts/gurobi_semantics_report.py r02"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402

from faster_amd import abi, capi, corridor  # noqa: E402
from test_gurobi_semantics import accounting  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
ctx = capi.Context(0)


def solve(p, f, par):
    ctx.set_params(par)
    return ctx.solve_batch(p, f)


sets = {
    "C2 safe N=6 P=1 (1024)": corridor.safe_batch(1024, seed=1)[:2],
    "C3 whole N=10 P<=4 (4096)": corridor.whole_batch(4096, seed=2, n_seg=10, p_choices=(2, 3, 4))[:2],
    "C4 whole N=10 P<=6 (8192)": corridor.whole_batch(8192, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))[:2],
}
try:
    from faster_amd import build as fb, frontend

    fb.build_frontend()
    pr, faces, _ = frontend.forest_batch(1024, seed=7, n_seg=15, max_poly=8)
    sets["C5 forest whole N=15 P<=8 (%d)" % len(pr)] = (pr, faces)
except Exception as e:  # the front-end needs g++
    print("C5 skipped:", e)
# C4 safe problems: the device hand-off of the whole solutions (R kept 0.05 m inside)
w, wf = sets["C4 whole N=10 P<=6 (8192)"]
pool = capi.Pool([0])
pool.set_pair_margin(0.05)
out = {}
for name, (pr, faces) in sets.items():
    out[name] = accounting(solve, pr, faces)
    print(name, json.dumps(out[name]))
out["_note"] = ("feas_tol 1e-9 -> 1e-6 and mip_gap 0 -> 1e-4 on the GPU path; flip_* = problems whose solved flag / winning factor / "
                "assignment changes; the reference's Gurobi defaults are FeasibilityTol 1e-6, MIPGap 1e-4 (SURVEY.md App. C)")
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_gurobi_semantics.json"), "w"), indent=1)
ctx.close()
pool.close()
