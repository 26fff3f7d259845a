#!/bin/bash
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
bash scripts/r4/ab.sh r4t default nolsv default nolsv
FASTERHIP_SO=build/libfasterhip_nolsv.so timeout 600 python scripts/records_bench.py 65536 2>&1 | grep "per cell\|16384" | cut -c1-200
