#!/bin/bash
# Collects the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   bash scripts/profile_round.sh r02
# kernel-trace/stats and every --pmc set are SEPARATE runs (gpurun refuses --pmc combined with sys/hip traces).
# Outputs land in gpurun_out/prof_<tag>/; scripts/summarize_profiles.py turns them into profiles/<tag>_*.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu --no-extra"
# 1. the driver's configuration (several steps in flight) and 2. one batch at a time (per-launch durations without overlap)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
tail -1 $OUT/stats.log | cut -c1-300
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_solo -o s -- $BENCH --inflight 1 --steps 32 > $OUT/stats_solo.log 2>&1
tail -1 $OUT/stats_solo.log | cut -c1-300
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
         "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $BENCH --inflight 1 --steps 8 --warmup 2 > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($C): rc=$?"
done
cd $R
python scripts/summarize_profiles.py $TAG
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/
