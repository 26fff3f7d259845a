#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3c; mkdir -p $O
cd $R
timeout 600 python scripts/r3/diag_n15.py > $O/diag_n15.txt 2>&1; cat $O/diag_n15.txt
FASTERHIP_SO=$R/build/variants/libfh_prof.so python scripts/phase_profile.py 8192 > $O/phase_w3.txt 2>&1; cat $O/phase_w3.txt
FASTERHIP_SO=$R/build/variants/libfh_prof2.so python scripts/phase_profile.py 8192 > $O/phase_w2.txt 2>&1; cat $O/phase_w2.txt
FASTERHIP_SO=$R/build/variants/libfh_w2.so python bench.py --no-cpu --no-extra > $O/bench_w2.json 2> $O/bench_w2.err; python -c "
import json;d=json.loads(open('$O/bench_w2.json').read().strip().splitlines()[-1]);print('w2',d['value'],d['ms_per_step'])"
FASTERHIP_SO=$R/build/variants/libfh_w2.so python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/bench_w2_solo.json 2>> $O/bench_w2.err; python -c "
import json;d=json.loads(open('$O/bench_w2_solo.json').read().strip().splitlines()[-1]);print('w2 solo',d['value'],d['ms_per_step'])"
