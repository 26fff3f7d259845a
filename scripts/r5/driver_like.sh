#!/bin/bash
# The driver's own command line (python bench.py --gpus 1 --steps 20 --warmup 5) with different numbers of pipelines in flight
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$1
for k in 8 10 12 14 20 8 10 12 14 20; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra --inflight $k > gpurun_out/$1/d_$k.json 2> gpurun_out/$1/d_$k.err
  python - $k gpurun_out/$1/d_$k.json <<'PY' | tee -a gpurun_out/$1/driver_like.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("steps 20 warmup 5, inflight %2s: %.2f M pairs/s, %.3f ms/step" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"]))
PY
done
