#!/bin/bash
# C5 numbers: bash scripts/r3/c5.sh [so]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
SO=${1:-faster_amd/libfasterhip.so}
FASTERHIP_SO=$R/$SO python bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --steps 16 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$SO C5: %.0f pairs/s %.1f ms/step; solved whole %.4f safe %.4f; iters/pair %.1f nodes whole %.1f safe %.1f' % (d['value'], d['ms_per_step'], c['whole_solved_frac'], c['safe_solved_frac'], c['mean_qp_iters_per_pair'], c['mean_bnb_nodes_whole'], c['mean_bnb_nodes_safe']))"
