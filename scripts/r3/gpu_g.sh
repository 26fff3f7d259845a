#!/bin/bash
# full GPU suite + solo bench + w2 comparison + phase profile
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3g; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -15 $O/pytest_gpu.txt
timeout 300 python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/bench_solo.json 2>> $O/bench.err; python -c "
import json;d=json.loads(open('$O/bench_solo.json').read().strip().splitlines()[-1]);print('solo',d['value'],d['ms_per_step'])"
FASTERHIP_SO=$R/build/variants/libfh_prof.so python scripts/phase_profile.py 8192 > $O/phase_w3.txt 2>&1; cat $O/phase_w3.txt
