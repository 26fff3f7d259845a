"""Check of the experimental FH_PARENT_BOUND build — children whose one-row dual bound at the parent already loses against the
incumbent are not visited (lead 0) —: results against the oracle, with and without work sharing, node counts.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -mllvm -sink-insts-to-avoid-spills -mllvm -disable-machine-licm \
          -DFH_PARENT_BOUND -I include -o build/libfasterhip_pb.so faster_amd/csrc/fh_capi.hip faster_amd/csrc/fh_pool.hip faster_amd/csrc/fh_map.hip
    FASTERHIP_SO=$PWD/build/libfasterhip_pb.so PYTHONPATH=. python scripts/r3/pb_check.py        (on the GPU box)
"""
import time
import numpy as np
import torch  # noqa: F401
from faster_amd import abi, capi, corridor
from oracle import oracle

print("library:", capi.SO_PATH, flush=True)


def check(name, pr, faces):
    ref = oracle.solve_batch(pr, faces)
    par = abi.default_params()
    a, b = capi.Context(0), capi.Context(0)
    a.set_params(par)
    got = a.solve_batch(pr, faces)
    par["share"] = 0
    b.set_params(par)
    alone = b.solve_batch(pr, faces)
    a.close(); b.close()
    flags = (alone["solved"] != ref["solved"]) | (alone["trials"] != ref["trials"]) | (alone["factor"] != ref["factor"]) | (alone["status"] != ref["status"])
    ok = (ref["solved"] == 1) & ~flags
    rel = np.abs(alone["cost"][ok] - ref["cost"][ok]) / np.maximum(np.abs(ref["cost"][ok]), 1e-2)
    cd = np.abs(alone["coeff"][ok] - ref["coeff"][ok]).reshape(ok.sum(), -1).max(axis=1)
    same = all(np.array_equal(alone[f], got[f]) for f in ("solved", "trials", "status", "factor", "dt", "cost", "coeff", "assign"))
    print("%s: %d problems, solved %.3f | flag mismatches %d, cost rel > 1e-7: %d (worst %.1e), coeff > 1e-6: %d (worst %.1e) | shared == alone bit for bit: %s | "
          "nodes device %.2f oracle %.2f, device <= oracle everywhere: %s, iterations device %.2f" % (
              name, len(pr), ref["solved"].mean(), int(flags.sum()), int((rel > 1e-7).sum()), rel.max() if ok.any() else 0, int((cd > 1e-6).sum()),
              cd.max() if ok.any() else 0, same, alone["nodes"].mean(), ref["nodes"].mean(), bool(np.all(alone["nodes"] <= ref["nodes"])), alone["qp_iters"].mean()), flush=True)


t = time.time()
pr, faces, _ = corridor.whole_batch(8192, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
check("C4 whole", pr, faces)
pr, faces, verts = corridor.safe_batch(2048, seed=610, n_seg=10, p_choices=(3, 4, 5))
faces = faces.copy(); faces["b"] -= 0.4
u = verts[:, 1] - verts[:, 0]; u /= np.linalg.norm(u, axis=1, keepdims=True)
pr["x0"][:, 3:6], pr["x0"][:, 6:9] = 4.8 * u, 2.0 * u
check("fast safe N=10", pr, faces)
pr, faces, _ = corridor.whole_batch(1024, seed=33, n_seg=15, p_choices=(4, 5, 6, 7, 8))
check("whole N=15", pr, faces)
pr, faces, _ = corridor.safe_batch(2048, seed=34, n_seg=6, p_choices=(1, 2, 3))
check("safe N=6", pr, faces)
print("%.1f s" % (time.time() - t))
