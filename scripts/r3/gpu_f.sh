#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
timeout 600 python scripts/r3/diag_n15.py > $O/diag_n15.txt 2>&1; grep -v "^   i" $O/diag_n15.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -8 $O/pytest_parity.txt
timeout 300 python bench.py --no-cpu --no-extra > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench',d['value'],d['ms_per_step'])"
