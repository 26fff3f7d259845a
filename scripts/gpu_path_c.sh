#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 PYTHONPATH=$PWD
timeout 600 python -u scripts/front_diag.py 4096 11 2>&1 | grep -v amdgpu.ids | tail -6
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py -q -m gpu -k "corridor_front_end or device_map or edge_cases or decomposition or closed_loop or forest" 2>&1 | tail -8
