#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONUNBUFFERED=1 PYTHONPATH=$R
timeout 600 python scripts/front_bench.py 65536 0 2>&1 | grep -v amdgpu | tail -1 | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "corridor_front_end or device_map or edge_cases" 2>&1 | tail -4
