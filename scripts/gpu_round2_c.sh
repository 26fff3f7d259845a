#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1
for cfg in "16 512" "16 2048" "4 512" "32 512" "16 128"; do
  set -- $cfg
  echo "===== min_nodes=$1 max_hungry=$2 : 32768"
  FH_DEBUG_MIN_NODES=$1 FH_DEBUG_MAX_HUNGRY=$2 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "^share1|stats|fused|==" | tail -7
done
echo "===== default 4096"
timeout 300 python -u scripts/share_diag.py 4096 2>&1 | grep -E "^share|stats|fused|==" | tail -12
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
