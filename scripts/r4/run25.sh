#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
FASTERHIP_SO=build/libfasterhip_dbgcodes.so ONLY_CFG=16 timeout 600 python tests/tools/path_sweep.py 2048 17 jps 2>&1 | grep -v ASTAR | tail -5
