#!/bin/bash
# the two SQ counter passes of the C5 kernel (after scripts/r3/final_pass.sh, which collects its stats and FETCH/WRITE passes), then the bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03_c5
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
PMC="python $R/bench.py --no-cpu --no-extra --workload c5 --pairs 65536 --inflight 1 --steps 4 --warmup 1"
i=2
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- $PMC > $OUT/pmc$i.log 2>&1
  echo "pmc$i: rc=$?"
done
cd $R
timeout 200 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 200 gpurun_out/bench_final.err
