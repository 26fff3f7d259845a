import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor
from oracle import oracle as orc
pr, faces, _ = corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))
ref = orc.solve_batch(pr, faces)
ctx = capi.Context(0)
par = abi.default_params(); par["share"] = 0; ctx.set_params(par)
for i in [113, 24]:
    one = pr[i:i + 1].copy()
    fo = int(one["face_begin"][0]); nf = int(one["face_off"][0][one["n_poly"][0]])
    fc = faces[fo:fo + nf].copy(); one["face_begin"] = 0
    one["f_init"] = ref["factor"][i]; one["f_final"] = ref["factor"][i]
    g1 = ctx.solve_batch(one, fc)
    print("i", i, "P", one["n_poly"][0], "solved", g1["solved"][0], "nodes", g1["nodes"][0], "oracle assign", list(ref["assign"][i][:15]))
    tr = g1["coeff"][0].reshape(-1)
    for k in range(min(int(g1["nodes"][0]), 32)):
        st_d, conf, qi, cost, sid, v = tr[6 * k:6 * k + 6]
        if int(st_d) % 10 == 1:
            src = int(sid // 1e10); rid = int(sid % 1e10)
            print("   node %2d depth %d INFEASIBLE conflict %s q %d iters %d src %d (1 const, 2 farkas) row kind %d t %d k %d f %d viol %.3e" % (k + 1, int(st_d) // 10, bin(int(conf)), int(qi) % 100, int(qi) // 100, src, rid >> 24, (rid >> 16) & 255, (rid >> 8) & 255, rid & 255, v))
        else:
            print("   node %2d depth %d st %d q %d iters %d cost %.6f" % (k + 1, int(st_d) // 10, int(st_d) % 10, int(qi) % 100, int(qi) // 100, cost))
