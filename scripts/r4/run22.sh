#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4w
timeout 900 python -m pytest tests -x -q -m gpu -k "decomp or corridor or safe or replan or polytope or front" > gpurun_out/r4w/test.txt 2>&1
tail -3 gpurun_out/r4w/test.txt
timeout 600 python scripts/replan_bench.py 2>&1 | grep -v ASTAR | tail -5 | cut -c1-1500
