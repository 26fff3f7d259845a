#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 PYTHONPATH=$PWD
timeout 300 python -u scripts/path_diag.py 2048 5 2>&1 | grep -v amdgpu.ids | tail -12
for w in 8 16 20; do
echo "== waves/CU $w"
FH_DEBUG_PLAN_WAVES_PER_CU=$w timeout 300 python -u scripts/path_diag.py 32768 7 2>&1 | grep -E "host|equal|differ" | head -8
done
