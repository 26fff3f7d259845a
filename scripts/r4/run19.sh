#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4s
timeout 900 python -m pytest tests -x -q -m gpu -k "jump_point or hashed or jps or path" > gpurun_out/r4s/test.txt 2>&1
tail -5 gpurun_out/r4s/test.txt
timeout 900 python scripts/records_bench.py 65536 gpurun_out/r4s/records.json 2>&1 | grep -v "ASTAR" | cut -c1-260 | tee gpurun_out/r4s/records.txt
FASTERHIP_SO=build/libfasterhip_jpsprof.so timeout 600 python scripts/jps_phase_profile.py 65536 0 2>&1 | grep -v ASTAR | tee gpurun_out/r4s/phases_dense.txt
