#!/bin/bash
# bench.py's timed region with 8 / 12 / 16 / 24 independent pipelines in flight:  bash scripts/r5/inflight.sh <tag>
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for k in 8 12 16 24 8 12 16; do
  timeout 300 python bench.py --no-cpu --no-extra --steps 96 --warmup 16 --inflight $k > gpurun_out/$1/inflight_$k.json 2> gpurun_out/$1/inflight_$k.err
  python - $k gpurun_out/$1/inflight_$k.json <<'PY' | tee -a gpurun_out/$1/inflight.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("inflight %2s: %.2f M pairs/s, %.3f ms/step" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"]))
PY
done
