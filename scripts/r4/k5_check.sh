#!/bin/bash
# the path-search tests of the GPU suite + the launch time of 65536 forest queries per kind of cell records (GPU box)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "jump_point or hashed or jps or path" 2>&1 | tail -2
timeout 900 python scripts/records_bench.py 65536 2>&1 | grep -v "ASTAR\|amdgpu" | python -c "
import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['map'][-12:], r['records'], round(r['launch_ms'],1), 'ms', r['queries_hit_the_limit'])"
