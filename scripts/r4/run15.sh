#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4o
timeout 600 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k hashed > gpurun_out/r4o/test.txt 2>&1
tail -15 gpurun_out/r4o/test.txt
timeout 900 python scripts/records_bench.py 65536 gpurun_out/r4o/records.json 2>&1 | grep -v "ASTAR" | tee gpurun_out/r4o/records.txt
