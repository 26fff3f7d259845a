#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4q
timeout 900 python -m pytest tests -x -q -m gpu -k "jump_point or hashed or jps or path" > gpurun_out/r4q/test.txt 2>&1
tail -5 gpurun_out/r4q/test.txt
timeout 900 python scripts/records_bench.py 65536 gpurun_out/r4q/records.json 2>&1 | grep -v "ASTAR" | tee gpurun_out/r4q/records.txt
