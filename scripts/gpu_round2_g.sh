#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in "" _noflops _nolook _neither; do for nf in 12 4; do
FASTERHIP_SO=$R/faster_amd/libfasterhip$v.so FH_DEBUG_MAX_HUNGRY=0 timeout 300 python bench.py --no-cpu --inflight $nf --steps 48 --warmup 8 > /tmp/r2.json 2>/tmp/r2.err; python - <<PY
import json
d=json.load(open("/tmp/r2.json")); print("variant '$v' no-share split inflight $nf: %.2f M pairs/s, %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
done; done
