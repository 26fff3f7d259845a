#!/usr/bin/env python3
"""Diff driver of the Gurobi tie-breaker.  TEST INFRASTRUCTURE.

Writes the problems of tests/golden (the reference's only fixture corridor with the survey's known answers KA-1..KA-4) plus a small
synthetic batch in the plain-number format of ref_driver.cpp, runs oracle/_ref/ref_driver (the UNTOUCHED solverGurobi.cpp, built by
build.sh where GUROBI_HOME exists), and compares with the CPU oracle — and with the GPU path when a device is present — under the
tolerances of SURVEY.md 8(c): `solved` and `factor_that_worked_` exact (cases whose feasibility margin lies between our 1e-9 and
Gurobi's FeasibilityTol 1e-6 are listed as marginal, not as failures), cost within 1e-4 relative (Gurobi's default MIPGap; our
exact optimum must not be larger), fillX positions within 1e-6.

Without the reference build it says so and exits 77: parity stays UNPINNED."""
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from faster_amd import abi, corridor  # noqa: E402
from oracle import oracle  # noqa: E402


def problems():
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "fixture_corridor.json")))
    ka = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    batches = [corridor.fixture_problem(fx, c["N"], c["vaj"], c["force_final"], c["polys"], c["x0"], c["xf"]) for c in ka["cases"].values()]
    pr, faces, _ = corridor.whole_batch(64, seed=2, n_seg=10, p_choices=(2, 3, 4))
    batches.append((pr, faces))
    pr, faces, _ = corridor.safe_batch(64, seed=1)
    batches.append((pr, faces))
    return corridor.concat(batches)


def write(path, pr, faces):
    with open(path, "w") as f:
        f.write("%d\n" % len(pr))
        for p in pr:
            f.write("%d %d %r %r %r %r %r %r %r\n" % (p["n_seg"], p["force_final_pos"], float(p["dc"]), float(p["v_max"]), float(p["a_max"]),
                                                      float(p["j_max"]), float(p["f_init"]), float(p["f_final"]), float(p["f_inc"])))
            f.write(" ".join(repr(float(v)) for v in p["x0"]) + "\n" + " ".join(repr(float(v)) for v in p["xf"]) + "\n%d\n" % p["n_poly"])
            for k in range(int(p["n_poly"])):
                f0, f1 = p["face_begin"] + p["face_off"][k], p["face_begin"] + p["face_off"][k + 1]
                f.write("%d\n" % (f1 - f0))
                for i in range(f0, f1):
                    f.write("%r %r %r %r\n" % (float(faces["a"][i][0]), float(faces["a"][i][1]), float(faces["a"][i][2]), float(faces["b"][i])))


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    if not os.path.exists(exe):
        rc = subprocess.call(["bash", os.path.join(HERE, "build.sh")])
        if rc != 0 or not os.path.exists(exe):
            print("diff_ref: no reference build (Gurobi absent): parity UNPINNED")
            return 77
    pr, faces = problems()
    inp = os.path.join(ROOT, "oracle", "_ref", "problems.txt")
    write(inp, pr, faces)
    ref = json.loads(subprocess.run([exe, inp], check=True, capture_output=True, text=True).stdout)
    ours = oracle.solve_batch(pr, faces)
    loose = abi.default_params()
    loose["feas_tol"] = 1e-6
    ours_loose = oracle.solve_batch(pr, faces, params=loose)
    bad = marginal = 0
    for i, r in enumerate(ref):
        same = r["solved"] == ours["solved"][i] and (not r["solved"] or r["factor"] == ours["factor"][i])
        if not same:
            if r["solved"] == ours_loose["solved"][i] and (not r["solved"] or r["factor"] == ours_loose["factor"][i]):
                marginal += 1
                print("marginal %d: feasibility margin between 1e-9 and Gurobi's FeasibilityTol" % i)
            else:
                bad += 1
                print("MISMATCH %d: reference solved %d factor %r, ours %d %r" % (i, r["solved"], r["factor"], ours["solved"][i], ours["factor"][i]))
            continue
        if r["solved"]:
            S = np.array(r["samples"])
            X = oracle.sample(pr[i], ours[i])
            jerks = np.unique(np.round(S[:-1, 3:6], 9), axis=0)
            cost_ref = float((jerks ** 2).sum())
            if not (ours["cost"][i] <= cost_ref * (1 + 1e-9) + 1e-9 and cost_ref - ours["cost"][i] <= 1e-4 * max(cost_ref, 1e-5)):
                bad += 1
                print("COST %d: reference %.12g ours %.12g" % (i, cost_ref, ours["cost"][i]))
            elif len(S) != len(X) or np.abs(S[:, :3] - X["pos"]).max() > 1e-6:
                print("note %d: same cost within the MIP gap, different trajectory (assignment within the gap?)" % i)
    print("diff_ref: %d problems, %d mismatches, %d marginal" % (len(ref), bad, marginal))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
