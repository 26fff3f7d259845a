#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4z
timeout 1500 python tests/tools/path_sweep.py 4096 48 jps 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r4z/path_sweep_jps.txt; tail -1 gpurun_out/r4z/path_sweep_jps.txt
timeout 1500 python tests/tools/path_sweep.py 4096 48 jps 32768 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r4z/path_sweep_jps_hashed.txt; tail -1 gpurun_out/r4z/path_sweep_jps_hashed.txt
