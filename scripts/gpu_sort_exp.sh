#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for s in "" 1; do
for f in 1 2 8; do
if [ -n "$s" ]; then export FH_DEBUG_NO_ORDER=1; else unset FH_DEBUG_NO_ORDER; fi
timeout 300 python bench.py --no-cpu --no-extra --inflight $f --steps 48 > /tmp/o.json 2>/tmp/o.err
python - <<PY
import json
d=json.load(open("/tmp/o.json")); print("no_order='$s' inflight $f: %.3f M pairs/s %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
done; done
unset FH_DEBUG_NO_ORDER
timeout 300 python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --steps 8 --warmup 2 > /tmp/o.json 2>/tmp/o.err
python - <<PY
import json
d=json.load(open("/tmp/o.json")); print("c5: %.3f M pairs/s %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
