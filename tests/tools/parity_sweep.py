"""Large randomized GPU-vs-oracle parity sweep (diagnostic; run on the GPU box). Reports every disagreement."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, capi, corridor
from oracle import oracle

ctx = capi.Context(0)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)
done, bad = 0, 0
worst_cost, worst_coeff = 0.0, 0.0
t0 = time.time()
cfg = 0
budget_s = float(sys.argv[2]) if len(sys.argv) > 2 else 240.0
while done < total and time.time() - t0 < budget_s:
    cfg += 1
    n_seg = int(rng.choice([3, 5, 6, 8, 10, 12, 15]))
    pmax = int(rng.integers(1, 9))
    pch = tuple(range(max(1, pmax - 2), pmax + 1))
    kw = dict(speed=float(rng.uniform(0.5, 4.5)), lateral=float(rng.uniform(0.0, 1.5)), acc0=float(rng.uniform(0, 3)),
              f_inc=float(rng.choice([0.5, 1.0, 1.0, 2.0])), v_max=float(rng.choice([3, 5])), a_max=float(rng.choice([3, 5])), j_max=float(rng.choice([5, 8])))
    n = 2048 if n_seg * pmax <= 60 else (512 if n_seg * pmax <= 80 else 96)  # dense N=15 corridors: exact enumeration is slow on both sides
    force = bool(rng.random() < 0.6)
    pr, faces, _ = corridor.make_batch(n, n_seg, pch, force, int(rng.integers(1 << 30)), **kw)
    if rng.random() < 0.3 and n_seg * pmax <= 40:   # tighter corridors: pull every face 0.3-0.8 m inwards (skipped for
        # many segments x many polytopes: mostly-infeasible MIQPs there need 1e4-1e5 nodes per trial — minutes of oracle time)
        faces = faces.copy(); faces["b"] -= rng.uniform(0.3, 0.8)
    ctx.set_sched(workgroups_per_cu=12 if cfg % 2 else 8)  # (both builds of the kernel: three / two wavefronts per SIMD, alternating)
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    flags = (got["solved"] != ref["solved"]) | (got["trials"] != ref["trials"]) | (got["factor"] != ref["factor"]) | (got["dt"] != ref["dt"]) | (got["status"] != ref["status"])
    ok = (ref["solved"] == 1) & ~flags
    rel = np.abs(got["cost"][ok] - ref["cost"][ok]) / np.maximum(np.abs(ref["cost"][ok]), 1e-9 / 1e-7)
    cdiff = np.abs(got["coeff"][ok] - ref["coeff"][ok]).reshape(ok.sum(), -1).max(axis=1) if ok.any() else np.zeros(0)
    nb = int(flags.sum() + (rel > 1e-7).sum() + (cdiff > 1e-6).sum())
    if nb:
        bad += nb
        for i in np.nonzero(flags)[0][:5]:
            print("  FLAG cfg", cfg, "i", i, "N", n_seg, "P", pr["n_poly"][i], "force", force, "gpu", (got["solved"][i], got["trials"][i], got["status"][i], got["cost"][i]),
                  "ref", (ref["solved"][i], ref["trials"][i], ref["status"][i], ref["cost"][i]))
        idx = np.nonzero(ok)[0]
        for j in np.nonzero((rel > 1e-7) | (cdiff > 1e-6))[0][:5]:
            i = idx[j]
            print("  NUM  cfg", cfg, "i", i, "N", n_seg, "P", pr["n_poly"][i], "cost", got["cost"][i], ref["cost"][i], "rel", rel[j], "coeff", cdiff[j], "nodes", got["nodes"][i], ref["nodes"][i])
    if ok.any():
        worst_cost = max(worst_cost, float(rel.max())); worst_coeff = max(worst_coeff, float(cdiff.max()))
    done += n
    print("cfg %3d N=%2d P<=%d force=%d solved %.2f | mismatches so far %d / %d | worst cost rel %.2e coeff %.2e | %.0fs" % (
        cfg, n_seg, pmax, force, ref["solved"].mean(), bad, done, worst_cost, worst_coeff, time.time() - t0), flush=True)
print("SWEEP DONE: %d problems, %d mismatches, worst cost rel %.3e, worst coeff abs %.3e" % (done, bad, worst_cost, worst_coeff))
