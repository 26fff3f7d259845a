#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4z
timeout 900 python -m pytest tests/test_gpu_round4.py -x -q -m gpu -k hashed 2>&1 | tail -5
timeout 1500 python tests/tools/path_sweep.py 4096 48 jps 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r4z/path_sweep_jps.txt; tail -1 gpurun_out/r4z/path_sweep_jps.txt
