#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
FASTERHIP_SO=build/libfasterhip_dbgcodes.so timeout 1500 python tests/tools/path_sweep.py 4096 48 jps 2>&1 | grep "at a limit\|DONE"
