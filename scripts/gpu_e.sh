#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_host_class.py tests/test_gpu_parity.py -q -m gpu -k "jps or decomposition" 2>&1 | tail -6
