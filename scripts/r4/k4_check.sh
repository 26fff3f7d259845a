#!/bin/bash
# decomposition / corridor tests of the GPU suite + the stage times of a faithful replan (GPU box)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "decomp or corridor or safe or replan or polytope or front" 2>&1 | tail -2
timeout 600 python scripts/replan_bench.py 2>&1 | grep -v ASTAR | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k:round(v,2) for k,v in d['stages_ms'].items()}, round(d['total_ms'],1), {k:round(v,2) for k,v in d['unknown_space_as_an_input']['stages_ms'].items()})"
