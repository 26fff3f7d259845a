#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONUNBUFFERED=1 PYTHONPATH=$R
timeout 600 python scripts/front_bench.py 65536 0 gpurun_out/r02_frontend_tmp.json > /dev/null 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r02_frontend_tmp.json'))['device']; print(d['stages_s'], d['pairs_per_s'])"
timeout 900 python -m pytest tests/test_gpu_round2.py tests/test_host_class.py tests/test_gpu_parity.py -q -m gpu -k "corridor_front_end or decomposition or forest or closed_loop" 2>&1 | tail -3
