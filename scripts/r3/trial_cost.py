"""What the factor trials that fail cost: C4's whole problems solved from f_init = 1 and from the factor that worked."""
import time
import numpy as np
import torch
from faster_amd import abi, capi, corridor

whole, faces, _ = corridor.whole_batch(32768, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
ctx = capi.Context(0)
ctx.set_params(abi.default_params())
res = ctx.solve_batch(whole, faces)
s = res["solved"] == 1
p2 = whole.copy()
p2["f_init"][s] = res["factor"][s]
for name, pr in (("all trials", whole), ("last trial only", p2)):
    best = 1e9
    for rep in range(4):
        t = time.perf_counter()
        r = ctx.solve_batch(pr, faces)
        best = min(best, time.perf_counter() - t)
    print("%s: %.2f ms (host pointers), trials %.2f nodes %.2f iters %.2f" % (name, best * 1e3, r["trials"].mean(), r["nodes"].mean(), r["qp_iters"].mean()), flush=True)
