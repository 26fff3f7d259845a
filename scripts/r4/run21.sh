#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
mkdir -p gpurun_out/r4v
FASTERHIP_SO=build/libfasterhip_jpsprof.so timeout 900 python scripts/jps_phase_profile.py 65536 32768 0.1 2>&1 | grep -v ASTAR | tee gpurun_out/r4v/phases_01_hashed.txt
