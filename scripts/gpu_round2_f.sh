#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r2f; mkdir -p $O
run() { # mh pipeline inflight
  FH_DEBUG_MAX_HUNGRY=$1 timeout 300 python bench.py --no-cpu --inflight $3 --pipeline $2 --steps 48 --warmup 8 > $O/b_$1_$2_$3.json 2> $O/b_$1_$2_$3.err
  python - <<PY
import json
try:
    d=json.load(open("$O/b_$1_$2_$3.json"))
    print("max_hungry $1 $2 inflight $3: %.2f M pairs/s, %.2f ms/step, avg launch %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]))
except Exception as e:
    print("$1 $2 $3 failed", e)
PY
}
run 0 split 12; run 0 fused 12
for mh in 16 64; do for pl in split fused; do for nf in 1 2 4 8 12; do run $mh $pl $nf; done; done; done
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
