#!/bin/bash
# Round-end pass on the GPU box, most important first (run through gpurun from the repo root):
#   GPU test suite, the default bench line, then the rocprofv3 evidence of the headline kernel and of the C5 kernel
#   (kernel-trace/stats and every --pmc set are SEPARATE runs).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "ASTAR ERROR" | tail -4
timeout 200 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 200 gpurun_out/bench_final.err
export TMPDIR=/tmp
cd /tmp
for TAG in r03 r03_c5; do
  OUT=$R/gpurun_out/prof_$TAG
  rm -rf $OUT; mkdir -p $OUT
  if [ $TAG = r03 ]; then
    BENCH="python $R/bench.py --no-cpu --no-extra"
    timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- $BENCH > $OUT/stats.log 2>&1
    SOLO="$BENCH --inflight 1 --steps 32"
    PMC="$BENCH --inflight 1 --steps 8 --warmup 2"
  else
    SOLO="python $R/bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --inflight 1 --steps 4 --warmup 1"
    PMC=$SOLO
  fi
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_solo -o s -- $SOLO > $OUT/stats_solo.log 2>&1
  i=0
  for C in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    if [ $TAG = r03_c5 ] && [ $i -gt 2 ]; then break; fi
    timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $PMC > $OUT/pmc$i.log 2>&1
    echo "$TAG pmc$i: rc=$?"
  done
  (cd $R && python scripts/summarize_profiles.py $TAG | tail -3 && mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* gpurun_out/profiles_$TAG/)
done
