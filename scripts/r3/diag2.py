"""Round-3 diagnostic: the trials the reduced-space kernel calls infeasible although the oracle solves them."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor
from oracle import oracle as orc

pr, faces, _ = corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))
ref = orc.solve_batch(pr, faces)
ctx = capi.Context(0)
par = abi.default_params(); par["share"] = 0; ctx.set_params(par)
got = ctx.solve_batch(pr, faces)
bad = np.nonzero((got["trials"] != ref["trials"]) | (got["solved"] != ref["solved"]))[0]
print("lib", capi.SO_PATH, "mismatches", list(bad))
for i in [3, 24, 79, 113, 114, 127]:
    one = pr[i:i + 1].copy()
    fo = int(one["face_begin"][0]); nf = int(one["face_off"][0][one["n_poly"][0]])
    fc = faces[fo:fo + nf].copy(); one["face_begin"] = 0
    one["f_init"] = ref["factor"][i]; one["f_final"] = ref["factor"][i]
    g1 = ctx.solve_batch(one, fc)
    o1 = orc.solve_batch(one, fc)
    pin = one.copy()
    a = ref["assign"][i]
    pins = 0
    for t in range(int(one["n_seg"][0])):
        pins |= (int(a[t]) + 1) << (4 * t)
    pin["pin"][0, 0] = pins & 0xffffffff; pin["pin"][0, 1] = pins >> 32
    g2 = ctx.solve_batch(pin, fc)
    o2 = orc.solve_batch(pin, fc)
    print("i", i, "factor", ref["factor"][i], "assign", list(a[:15]))
    print("   single trial: gpu solved/status/nodes/iters", g1["solved"][0], g1["status"][0], g1["nodes"][0], g1["qp_iters"][0], "cost", g1["cost"][0], "| oracle", o1["solved"][0], o1["cost"][0], "assign", list(g1["assign"][0][:15]))
    print("   pinned      : gpu solved/status/nodes/iters", g2["solved"][0], g2["status"][0], g2["nodes"][0], g2["qp_iters"][0], "cost", g2["cost"][0], "| oracle", o2["solved"][0], o2["cost"][0])
