#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r2p; mkdir -p $O
for nf in 1 8; do
timeout 900 python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --inflight $nf --steps 8 --warmup 2 > $O/c5_$nf.json 2> $O/c5_$nf.err; python - <<PY
import json
try:
    d=json.load(open("$O/c5_$nf.json")); print("C5 65536 fused inflight $nf: %.1f k pairs/s, %.1f ms/step" % (d["value"]/1e3, d["ms_per_step"]), json.dumps(d["config"])[:600])
except Exception as e: print("c5 $nf failed", e, open("$O/c5_$nf.err").read()[-500:])
PY
done
timeout 900 python bench.py --no-cpu --no-extra --no-share --workload c5 --pairs 65536 --inflight 8 --steps 8 --warmup 2 > $O/c5_ns.json 2> $O/c5_ns.err; python - <<PY
import json
d=json.load(open("$O/c5_ns.json")); print("C5 65536 no-share inflight 8: %.1f k pairs/s, %.1f ms/step" % (d["value"]/1e3, d["ms_per_step"]))
PY
