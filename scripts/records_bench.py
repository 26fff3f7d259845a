"""Jump point search on the device: per-cell records vs hashed records (fh_map_set_records).  Launch time of the path search alone,
queries that hit the limit, workspace bytes.   usage: records_bench.py [n_queries] [out.json]   (GPU box; PYTHONPATH = repo root)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import capi, frontend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
out = sys.argv[2] if len(sys.argv) > 2 else None
torch.cuda.init()
rows = []
for name, res, size, infl in (("forest 20x20x3 m, 0.2 m cells", 0.2, None, 0.3), ("forest 20x20x3 m, 0.1 m cells", 0.1, None, 0.3)):
    cloud, cells, center, starts, goals = frontend.forest_queries(n, 5)
    if res != 0.2:
        cells = tuple(int(round(c * 0.2 / res)) for c in cells)
    m = capi.Map(0)
    m.set_search("jps")
    m.read(cloud, cells, res, center, 0.0, 3.0, infl)
    dev = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).cuda()
    d_s, d_g = dev(starts, np.float64), dev(goals, np.float64)
    mp = 64
    d_p = torch.empty((n, mp, 3), dtype=torch.float64, device="cuda")
    d_n = torch.empty(n, dtype=torch.int32, device="cuda")
    d_e = torch.empty(n, dtype=torch.int64, device="cuda")
    ref = None
    for slots in (0, 4096, 8192, 16384, 32768):
        m.set_records(slots)
        best = 1e9
        for rep in range(4):
            torch.cuda.synchronize()
            t = time.perf_counter()
            m.plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), n, mp, d_p.data_ptr(), d_n.data_ptr(), d_e.data_ptr())
            m.sync()
            dt = time.perf_counter() - t
            if rep > 0:
                best = min(best, dt)
        npts, ex, paths = d_n.cpu().numpy(), d_e.cpu().numpy(), d_p.cpu().numpy()
        if ref is None:
            ref = (npts.copy(), ex.copy(), paths.copy())
        over = npts == -2
        ok = ~over
        same = bool(np.array_equal(npts[ok], ref[0][ok]) and np.array_equal(ex[ok], ref[1][ok]) and
                    all(np.array_equal(paths[i, :npts[i]], ref[2][i, :npts[i]]) for i in np.nonzero(ok & (npts > 0))[0][::16]))
        row = {"map": name, "cells": int(np.prod(cells)), "queries": n, "records": "per cell" if slots == 0 else "%d hashed slots" % slots,
               "launch_ms": best * 1e3, "queries_hit_the_limit": int(over.sum()), "same_as_per_cell_records": same,
               "workspace_bytes": m.workspace_bytes(), "pops": int(ex.sum()), "max_pops": int(ex.max())}
        rows.append(row)
        print(json.dumps(row), flush=True)
    m.close()
if out:
    json.dump(rows, open(out, "w"), indent=1)
