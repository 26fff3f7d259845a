#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONUNBUFFERED=1 PYTHONPATH=$R
timeout 600 python scripts/front_bench.py 65536 8192 gpurun_out/r02_frontend.json 2>&1 | grep -v amdgpu | tail -1 | cut -c200-1200
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_host_class.py tests/test_gpu_parity.py -q -m gpu -k "corridor_front_end or device_map or edge_cases or jps or forest or closed_loop" 2>&1 | tail -4
