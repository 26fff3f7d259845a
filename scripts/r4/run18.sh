#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4r
FASTERHIP_SO=build/libfasterhip_jpsprof.so timeout 600 python scripts/jps_phase_profile.py 65536 0 2>&1 | grep -v ASTAR | tee gpurun_out/r4r/phases_dense.txt

