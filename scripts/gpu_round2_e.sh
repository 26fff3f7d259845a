#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1
echo "===== margin 0.05, 32768"
PAIR_MARGIN=0.05 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "^share|hardest|solved|fused|==" | tail -16
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
