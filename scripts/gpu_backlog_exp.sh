#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { timeout 300 python bench.py --no-cpu --no-extra --inflight $1 --steps 64 > /tmp/o.json 2>/tmp/o.err
python - <<PY
import json
d=json.load(open("/tmp/o.json")); print("factor=$FH_DEBUG_GIANT_FACTOR backlog=$FH_DEBUG_BACKLOG inflight $1: %.3f M pairs/s %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
}
for f in 6 4 3 2; do export FH_DEBUG_GIANT_FACTOR=$f; run 1; done
export FH_DEBUG_GIANT_FACTOR=3 FH_DEBUG_BACKLOG=64; run 1; run 8
