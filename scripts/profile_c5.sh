#!/bin/bash
# rocprofv3 evidence for the C5 kernel (fh::solve_kernel<15, true, 2>): kernel stats and PMC passes of bench.py --workload c5
set -u
TAG=${1:-r04_c5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu --no-extra --workload c5 --c5-rule reference --pairs 65536 --inflight 1 --steps 4 --warmup 1"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_solo -o s -- $BENCH > $OUT/stats_solo.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($C): rc=$?"
done
cd $R
python scripts/summarize_profiles.py $TAG
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/
