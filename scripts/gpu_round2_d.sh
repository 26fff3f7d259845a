#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1
export FASTERHIP_SO=$R/faster_amd/libfasterhip_prof.so
for cfg in "16 512" "32 512"; do
  set -- $cfg
  echo "===== PROFILE BUILD min_nodes=$1 max_hungry=$2 : 32768"
  FH_DEBUG_MIN_NODES=$1 FH_DEBUG_MAX_HUNGRY=$2 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "^share1|stats|profile|frame|fused|==" | tail -9
done
echo "===== PROFILE 4096"
timeout 300 python -u scripts/share_diag.py 4096 2>&1 | grep -E "^share1|stats|profile|frame|fused|==" | tail -9
unset FASTERHIP_SO
echo "===== PRODUCT BUILD 32768 (min 16/32/8)"
for mn in 16 32 8; do FH_DEBUG_MIN_NODES=$mn timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "^share1|fused|==" | tail -4; done
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
