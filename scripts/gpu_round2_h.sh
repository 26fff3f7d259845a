#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r2h; mkdir -p $O
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^Activated\|^factor_initial" | tail -25
echo "== bench default"; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo rc=$?; tail -3 $O/bench.err; python - <<PY
import json
d=json.load(open("$O/bench.json"))
print(json.dumps({k:d[k] for k in ("value","ms_per_step")}))
print(json.dumps(d["roofline"], indent=1)[:3000])
print(json.dumps(d.get("e2e_with_copies")), json.dumps(d.get("cpu_baseline")))
print(json.dumps(d["config"], indent=1)[:1500])
PY
