"""Corridor front-end of BASELINE config 5 on the device vs the CPU front-end: stage times, pairs/s, bytes and expansions.
usage: front_bench.py [n_pairs] [host_sample] [out.json]   (run on the GPU box; PYTHONPATH = repo root)"""
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import capi, frontend  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
host_n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
out = sys.argv[3] if len(sys.argv) > 3 else None
seed = 5
ctx, vmap = capi.Context(0), capi.Map(0)
cells = 117 * 117 * 14
res = {
    "workload": "BASELINE config 5 front-end: %d start/goal pairs in one random forest (20x20x3 m, 117x117x14 cells of 0.2 m), "
                "path search + createMoreVertexes/deleteVertexes + ellipsoid decomposition (<=8 polytopes)" % n,
}
for search in ("jps", "astar"):
    frontend.forest_batch(256, seed, front="device", ctx=ctx, vmap=vmap, search=search)  # allocations, first-touch
    best = None
    for rep in range(3):
        pr, fc, info = frontend.forest_batch(n, seed, front="device", ctx=ctx, vmap=vmap, search=search)
        tm = info["front_timing"]
        tot = tm["map_s"] + tm["path_search_s"] + tm["decomposition_s"]
        if best is None or tot < best[0]:
            best = (tot, tm, len(pr))
    tot, tm, kept = best
    waves = 256 * (16 if search == "jps" else 12)
    res["device_" + search] = {"pairs": n, "kept": kept, "seconds": tot, "pairs_per_s": n / tot, "stages_s": tm,
                               "expansions_per_s": tm["expansions"] / tm["path_search_s"], "mean_expansions_per_query": tm["expansions"] / n,
                               "wavefronts": waves, "workspace_bytes_per_wavefront": cells * 16 + 2048 * 768,
                               **({"jump_table_bytes": cells * 64, "note": "jump point search in jps3d's own order (FASTER's exact vertex "
                                   "lists); path_search_s includes building the jump tables of the map (three launches)"} if search == "jps" else
                                  {"note": "A* with a total order of its own (an optimal path)"})}
res["device"] = res["device_jps"]
if host_n > 0:
    t = time.perf_counter()
    hp, hf, hi = frontend.forest_batch(host_n, seed, search="jps")
    th = time.perf_counter() - t
    res["host"] = {"pairs": host_n, "seconds": th, "pairs_per_s": host_n / th, "threads": len(os.sched_getaffinity(0)),
                   "note": "faster_amd/host/corridor_frontend.cpp, OpenMP over pairs (plan_path_jps), incl. the numpy assembly of the problem records"}
    res["speedup_vs_host"] = res["device"]["pairs_per_s"] / res["host"]["pairs_per_s"]
print(json.dumps(res))
if out:
    json.dump(res, open(out, "w"), indent=1)
