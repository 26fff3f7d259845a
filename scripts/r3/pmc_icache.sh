#!/bin/bash
# (counter sets beyond the ones used in DESIGN.md section 4 have hung rocprofv3 on this pool — TCP_UTCL1_* / TCP_TCC_READ_REQ_LATENCY cost a 15-minute timeout: do not pass them)
# instruction-cache counters of the solve kernel (one launch at a time)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_icache
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python $R/bench.py --no-cpu --no-extra --inflight 1 --steps 8 --warmup 2"
timeout 600 rocprofv3 --pmc ${PMC:-SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH} --kernel-trace --output-format csv -d $OUT/p1 -o p -- $BENCH > $OUT/p1.log 2>&1
echo rc=$?
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p1/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    if "solve_kernel" in k:
        print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
