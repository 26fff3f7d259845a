#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { timeout 300 python bench.py --no-cpu --no-extra --inflight $1 --steps 48 > /tmp/o.json 2>/tmp/o.err
python - <<PY
import json
d=json.load(open("/tmp/o.json")); print("backlog=$FH_DEBUG_BACKLOG giant=$FH_DEBUG_GIANT inflight $1: %.3f M pairs/s %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
}
run 1; run 8
for b in 4 16; do for g in 256 1024; do export FH_DEBUG_BACKLOG=$b FH_DEBUG_GIANT=$g; run 1; run 8; done; done
