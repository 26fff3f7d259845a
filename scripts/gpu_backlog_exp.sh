#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { timeout 300 python bench.py --no-cpu --no-extra --inflight $1 --steps 64 > /tmp/o.json 2>/tmp/o.err
python - <<PY
import json
d=json.load(open("/tmp/o.json")); print("inflight $1: %.3f M pairs/s %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
}
run 1; run 2; run 8; run 8; run 12
timeout 300 python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --steps 8 --warmup 2 > /tmp/o.json 2>/tmp/o.err; python -c "
import json; d=json.load(open('/tmp/o.json')); print('c5', d['value'], d['ms_per_step'])"
