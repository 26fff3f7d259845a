#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for cfg in "0 64" "8 64" "8 128" "32 48" "32 96"; do set -- $cfg; for nf in 1 4; do FH_DEBUG_BACKLOG=$1 FH_DEBUG_GIANT=$2 timeout 300 python bench.py --no-cpu --no-extra --inflight $nf --steps 48 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("backlog $1 giant $2 fused inflight $nf: %.2f M pairs/s, %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]), d["config"]["share_stats_last_launch"])
PY
done; done
