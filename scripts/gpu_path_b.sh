#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 PYTHONPATH=$PWD
timeout 600 python -m pytest tests/test_gpu_round2.py -q -m gpu -k "device_map or edge_cases" 2>&1 | tail -15
