#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "" _look8 _look16; do for nf in 1 8; do
FASTERHIP_SO=$R/faster_amd/libfasterhip$v.so timeout 300 python bench.py --no-cpu --no-extra --inflight $nf --steps 64 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("variant '$v' fused inflight $nf: %.2f M pairs/s, %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
done; done
