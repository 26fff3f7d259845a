#!/bin/bash
# A/B bench on one box: bash scripts/r3/ab.sh <out> <so1> <so2> ... (3 alternating rounds, driver configuration and solo)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O; shift
cd $R
for round in 1 2 3; do
  for so in "$@"; do
    n=$(basename $so .so)
    FASTERHIP_SO=$R/$so python bench.py --no-cpu --no-extra > $O/b_${n}_$round.json 2>> $O/err.txt
    FASTERHIP_SO=$R/$so python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/s_${n}_$round.json 2>> $O/err.txt
    python - <<PY
import json
d=json.loads(open('$O/b_${n}_$round.json').read().strip().splitlines()[-1]); s=json.loads(open('$O/s_${n}_$round.json').read().strip().splitlines()[-1])
print('%-28s round $round: %.3f M pairs/s (%.3f ms/step)   solo %.3f M (%.3f ms)' % ('$n', d['value']/1e6, d['ms_per_step'], s['value']/1e6, s['ms_per_step']))
PY
  done
done
