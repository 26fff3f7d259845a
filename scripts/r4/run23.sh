#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
for k in 0 1 3; do
  echo stop_after=$k; FHD_STOP_AFTER=$k FASTERHIP_SO=build/libfasterhip_k4exp.so timeout 600 python scripts/replan_bench.py 2>&1 | grep -v ASTAR | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:round(v,2) for k,v in d['stages_ms'].items()}, {k:round(v,2) for k,v in d['unknown_space_as_an_input']['stages_ms'].items()})"
done
