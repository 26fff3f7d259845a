import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor
from oracle import oracle as orc
pr, faces, _ = corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))
ref = orc.solve_batch(pr, faces)
ctx = capi.Context(0)
par = abi.default_params(); par["share"] = 0; ctx.set_params(par)
for i in [113]:
    one = pr[i:i + 1].copy()
    fo = int(one["face_begin"][0]); nf = int(one["face_off"][0][one["n_poly"][0]])
    fc = faces[fo:fo + nf].copy(); one["face_begin"] = 0
    one["f_init"] = ref["factor"][i]; one["f_final"] = ref["factor"][i]
    g1 = ctx.solve_batch(one, fc)
    print("i", i, "P", one["n_poly"][0], "solved", g1["solved"][0], "nodes", g1["nodes"][0], "iters", g1["qp_iters"][0], "dt", g1["dt"][0])
    tr = g1["coeff"][0].reshape(-1)[:192].reshape(32, 6)
    tr = tr[np.argsort(tr[:, 0])]
    for r in tr:
        rid = int(r[1])
        print("   it %3d row kind %d t %2d k %d f %3d  vp %.6e zz %.6e gg %.6e zz/gg %.3e  %s %.6e" % (int(r[0]), rid >> 24, (rid >> 16) & 255, (rid >> 8) & 255, rid & 255, r[2], r[3], r[4], r[3] / max(r[4], 1e-300), "DROP t1" if r[5] < 0 else "add t2", abs(r[5])))
