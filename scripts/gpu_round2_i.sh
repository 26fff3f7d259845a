#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
python -u - <<'PY'
import sys, time
sys.path.insert(0, "tests")
import numpy as np, torch
from faster_amd import abi, capi, corridor
from test_gpu_round2 import hard_problems
for share in (1, 0):
    c = capi.Context(0); par = abi.default_params(); par["share"] = share; c.set_params(par)
    easy, ef, _ = corridor.whole_batch(64, seed=5); c.solve_batch(easy, ef)
    for n, seed in ((1, 78), (1, 79), (1, 80), (1, 81), (6, 77), (64, 77)):
        pr, faces = hard_problems(n=n, seed=seed)
        t = time.perf_counter(); r = c.solve_batch(pr, faces); dt = time.perf_counter() - t
        print("share %d n %d seed %d: %.1f ms, solved %d, nodes max %d, trials max %d" % (share, n, seed, 1e3 * dt, r["solved"].sum(), r["nodes"].max(), r["trials"].max()), flush=True)
    c.close()
PY
