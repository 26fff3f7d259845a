#!/bin/bash
# rocprofv3 kernel stats of the corridor front-end (both path searches) + its bench record
set -u
TAG=${1:-r03_front}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
timeout 600 python $R/scripts/front_bench.py 65536 4096 $OUT/front_end.json > $OUT/front.log 2>&1
tail -1 $OUT/front.log | cut -c1-600
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python $R/scripts/front_bench.py 65536 0 > $OUT/stats.log 2>&1
ls $OUT/stats/* | head
