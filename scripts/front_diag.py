"""Corridor front-end host vs device on the C5 forest: corridors (counts, rows), solver results, timings.  usage: front_diag.py [n] [seed]"""
import sys
import time

import numpy as np
import torch  # noqa: F401

from faster_amd import abi, capi, frontend

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
t = time.time()
hp, hf, hi = frontend.forest_batch(n, seed)
t_host = time.time() - t
ctx, vmap = capi.Context(0), capi.Map(0)
frontend.forest_batch(256, seed, front="device", ctx=ctx, vmap=vmap)  # warm-up (allocations)
t = time.time()
dp, df, di = frontend.forest_batch(n, seed, front="device", ctx=ctx, vmap=vmap)
t_dev = time.time() - t
print("host front-end %.2f s (%.0f pairs/s incl. python assembly), device %.2f s; device stages:" % (t_host, n / t_host, t_dev), di["front_timing"])
tm = di["front_timing"]
print("device front-end kernels only: %.0f pairs/s" % (n / (tm["map_s"] + tm["path_search_s"] + tm["decomposition_s"])))
print("kept", len(hp), len(dp), "same pairs kept:", np.array_equal(hi["kept"], di["kept"]))
if np.array_equal(hi["kept"], di["kept"]):
    print("n_poly equal:", np.array_equal(hp["n_poly"], dp["n_poly"]), " face_off equal:", np.array_equal(hp["face_off"], dp["face_off"]))
    same = np.all(hp["face_off"] == dp["face_off"], axis=1) & (hp["n_poly"] == dp["n_poly"])
    print("pairs with identical polytope sizes: %d / %d" % (same.sum(), len(same)))
    key = lambda M: M[np.lexsort(np.round(M, 6).T[::-1])]
    worst, bad = 0.0, 0
    for i in np.nonzero(same)[0]:
        for p in range(hp["n_poly"][i]):
            a0, a1 = hp["face_off"][i][p], hp["face_off"][i][p + 1]
            H = np.column_stack([hf["a"][hp["face_begin"][i] + a0: hp["face_begin"][i] + a1], hf["b"][hp["face_begin"][i] + a0: hp["face_begin"][i] + a1]])
            D = np.column_stack([df["a"][dp["face_begin"][i] + a0: dp["face_begin"][i] + a1], df["b"][dp["face_begin"][i] + a0: dp["face_begin"][i] + a1]])
            dm = np.abs(H[:, None, :] - D[None, :, :]).max(axis=2)   # rows matched to their nearest counterpart
            d = max(dm.min(axis=1).max(), dm.min(axis=0).max())
            worst = max(worst, d)
            if d > 1e-9 and bad < 2:
                np.set_printoptions(precision=6, suppress=True, linewidth=200)
                print('pair', i, 'polytope', p, 'host rows'); print(H); print('device rows'); print(D)
            bad += d > 1e-9
    print("worst row difference %.3g, polytopes beyond 1e-9: %d" % (worst, bad))
    print("x0/xf equal:", np.array_equal(hp["x0"], dp["x0"]), np.abs(hp["xf"] - dp["xf"]).max())
    rh, rd = ctx.solve_batch(hp, hf), ctx.solve_batch(dp, df)
    print("solved host-corridors %.4f device-corridors %.4f, flags equal %d / %d" % (rh["solved"].mean(), rd["solved"].mean(), (rh["solved"] == rd["solved"]).sum(), len(rh)))
    both = (rh["solved"] == 1) & (rd["solved"] == 1)
    print("cost rel diff max", np.max(np.abs(rh["cost"][both] - rd["cost"][both]) / np.maximum(rh["cost"][both], 1e-12)))
