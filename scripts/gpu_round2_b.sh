#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
export PYTHONUNBUFFERED=1
echo "== 1: pytest synthetic, default"; timeout 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "dt_initial or known or synthetic" 2>&1 | grep -v "^  File" | tail -8
echo "== 2: pytest synthetic, no hungry cap"; FH_DEBUG_MAX_HUNGRY=1000000 timeout 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "dt_initial or known or synthetic" 2>&1 | grep -v "^  File" | tail -8
echo "== 3: diag 4096"; timeout 300 python -u scripts/share_diag.py 4096 2>&1 | tail -30
