#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -4 $O/pytest_parity.txt
timeout 300 python bench.py --no-cpu --no-extra > $O/bench.json 2> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print('bench',d['value'],d['ms_per_step'])"
timeout 300 python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/bench_solo.json 2>> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench_solo.json').read().strip().splitlines()[-1]);print('solo',d['value'],d['ms_per_step'])"
FASTERHIP_SO=$R/build/variants/libfh_prof.so python scripts/phase_profile.py 8192 > $O/phase.txt 2>&1; cat $O/phase.txt
