"""Where a wavefront of the jump point search spends its cycles (FHP_PROFILE build of the library: bash scripts/build_variant.sh
jpsprof -DFHP_PROFILE; run with FASTERHIP_SO=build/libfasterhip_jpsprof.so).  One launch per phase: `expansions` receives the cycles
of that phase.   usage: jps_phase_profile.py [n_queries] [slots] [cell size, default 0.2]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import capi, frontend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 0
res = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
torch.cuda.init()
cloud, cells, center, starts, goals = frontend.forest_queries(n, 5)
m = capi.Map(0)
m.set_search("jps")
m.set_records(slots)
if res != 0.2:
    cells = tuple(int(round(c * 0.2 / res)) for c in cells)
m.read(cloud, cells, res, center, 0.0, 3.0, 0.3)
dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
d_s, d_g = dev(starts, np.float64), dev(goals, np.float64)
mp = 64
d_p = torch.empty((n, mp, 3), dtype=torch.float64, device="cuda")
d_n = torch.empty(n, dtype=torch.int32, device="cuda")
d_e = torch.empty(n, dtype=torch.int64, device="cuda")


def launch(slot):
    os.environ["FHP_PROFILE_SLOT"] = str(slot)
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t = time.perf_counter()
        m.plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), n, mp, d_p.data_ptr(), d_n.data_ptr(), d_e.data_ptr())
        m.sync()
        best = min(best, time.perf_counter() - t)
    return best, d_e.cpu().numpy().astype(np.float64)


ms, pops = launch(-1)
total_pops = pops.sum()
ms7, whole = launch(7)
names = ["pop + candidate addresses", "heap sift-down", "candidates settled from the tables", "cell-by-cell jumps",
         "move costs + wait for the cell records", "relaxation (records written)", "heap pushes / increases"]
print("%d queries, %d pops (%.1f per query), launch %.1f ms (profile build: %.1f ms); cycles per query %.0f = %.0f per pop" %
      (n, total_pops, total_pops / n, ms * 1e3, ms7 * 1e3, whole.mean(), whole.sum() / total_pops))
acc = 0.0
for k, name in enumerate(names):
    _, c = launch(k)
    acc += c.sum()
    print("  %-42s %5.1f %%   %7.0f cycles per pop" % (name, 100 * c.sum() / whole.sum(), c.sum() / total_pops))
_, c8 = launch(8)
_, c9 = launch(9)
_, c10 = launch(10)
_, c11 = launch(11)
print("  sift-down: heap in LDS %.0f cycles per such pop (%.1f %% of the pops), heap deeper than LDS %.0f cycles per such pop (%.1f %%); mean heap size %.0f" %
      (c8.sum() / max(total_pops - c10.sum(), 1), 100 * (1 - c10.sum() / total_pops), c9.sum() / max(c10.sum(), 1), 100 * c10.sum() / total_pops, c11.sum() / total_pops))
print("  %-42s %5.1f %%   %7.0f cycles per pop" % ("outside the loop (boxes, path, clean-up)", 100 * (1 - acc / whole.sum()), (whole.sum() - acc) / total_pops))
m.close()
