#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1
PAIR_MARGIN=0.05 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "^share|solved|fused|==" | tail -12
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for nf in 1 2 8; do timeout 300 python bench.py --no-cpu --no-extra --inflight $nf --steps 64 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("fused inflight $nf: %.2f M pairs/s, %.2f ms/step iters/pair %.1f" % (d["value"]/1e6, d["ms_per_step"], d["config"]["mean_qp_iters_per_pair"]))
PY
done
timeout 300 python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --steps 8 --warmup 2 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("C5: %.1f k pairs/s, %.1f ms/step iters/pair %.1f" % (d["value"]/1e3, d["ms_per_step"], d["config"]["mean_qp_iters_per_pair"]))
PY
timeout 400 python tests/tools/parity_sweep.py 150000 200 2>&1 | tail -2
