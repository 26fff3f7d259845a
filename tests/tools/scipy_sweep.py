"""The oracle against an independent SciPy (SLSQP) solve of the UNREDUCED 12N-coefficient model, written row by row as solverGurobi.cpp
adds them (oracle/py_model.py), under the oracle's assignment: cost and polynomial coefficients.  CPU only.
    PYTHONPATH=. python tests/tools/scipy_sweep.py [problems_per_config]"""
import sys
import time

import numpy as np

from faster_amd import corridor
from oracle import oracle, py_model


def polys_of(pr, faces):
    out = []
    fb = int(pr["face_begin"])
    for p in range(int(pr["n_poly"])):
        f0, f1 = fb + pr["face_off"][p], fb + pr["face_off"][p + 1]
        out.append((faces["a"][f0:f1].copy(), faces["b"][f0:f1].copy()))
    return out


n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
t0 = time.time()
tot = bad = fail = 0
worst_c = worst_x = 0.0
for cfg, (n_seg, pch, force) in enumerate([(6, (1, 2, 3), True), (6, (1, 2, 3), False), (8, (2, 3, 4), True), (10, (2, 4, 6), True), (10, (2, 3), False), (5, (2, 3), True)]):
    mk = corridor.whole_batch if force else corridor.safe_batch
    pr, faces, verts = mk(n, seed=900 + cfg, n_seg=n_seg, p_choices=pch)
    if not force:  # fast safe problems: the branching rule of DESIGN.md 4a is at work
        faces = faces.copy()
        faces["b"] -= 0.3
        u = verts[:, 1] - verts[:, 0]
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        pr["x0"][:, 3:6], pr["x0"][:, 6:9] = 4.5 * u, 1.5 * u
    res = oracle.solve_batch(pr, faces)
    for i in np.nonzero(res["solved"])[0]:
        p, r = pr[i], res[i]
        N = int(p["n_seg"])
        s = py_model.solve_fixed(N, float(r["dt"]), p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]),
                                 bool(p["force_final_pos"]), polys_of(p, faces), [int(a) for a in r["assign"][:N]])
        tot += 1
        if s is None:
            fail += 1
            continue
        dc = abs(s[0] - r["cost"]) / max(abs(r["cost"]), 1e-2)
        dx = float(np.abs(s[1] - r["coeff"][:N]).max())
        worst_c, worst_x = max(worst_c, dc), max(worst_x, dx)
        if dc > 1e-6 or dx > 5e-5:
            bad += 1
            print("  DIFF cfg %d problem %d: cost %.9g vs %.9g, coeff %.2e" % (cfg, i, s[0], r["cost"], dx), flush=True)
    print("cfg %d N=%d P in %s force=%d: solved %.2f | compared so far %d, SLSQP failed to converge %d, differences %d | worst cost rel %.1e coeff %.1e | %ds"
          % (cfg, n_seg, pch, force, res["solved"].mean(), tot, fail, bad, worst_c, worst_x, time.time() - t0), flush=True)
print("SCIPY SWEEP DONE: %d solved problems compared, %d SLSQP failures, %d differences (cost rel > 1e-6 or coefficient > 5e-5), worst cost rel %.2e, worst coefficient %.2e"
      % (tot, fail, bad, worst_c, worst_x))
