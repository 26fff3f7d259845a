#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1
PAIR_MARGIN=0.05 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "fused|==" | tail -5
for pl in fused split; do for nf in 1 4 12; do timeout 300 python bench.py --no-cpu --no-extra --pipeline $pl --inflight $nf --steps 48 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("$pl inflight $nf: %.2f M pairs/s, %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
done; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
