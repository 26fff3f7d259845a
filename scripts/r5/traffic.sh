#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the solve kernel of library variants:  bash scripts/r5/traffic.sh <tag> <variant>...
set -u
R=$GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
for v in "$@"; do
  so=$R/build/libfasterhip_$v.so; [ "$v" = default ] && so=$R/faster_amd/libfasterhip.so
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=$R/gpurun_out/$TAG/$v/$C; mkdir -p $OUT
    ( cd /tmp && FASTERHIP_SO=$so timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -o p -- python $R/bench.py --no-cpu --no-extra --inflight 1 --steps 8 --warmup 2 > $OUT/log.txt 2>&1 )
  done
  python - $R/gpurun_out/$TAG/$v $v <<'PY'
import csv, glob, sys, collections
tot = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(float)
    for f in glob.glob(sys.argv[1] + "/" + C + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "solve_kernel" in r["Kernel_Name"] and r["Counter_Name"] == C:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    tot[C] = sum(per.values()) / max(1, len(per))
print("%-10s fetch x2 %.1f MB  write %.1f MB  total %.1f MB per launch" % (sys.argv[2], tot["FETCH_SIZE"] * 2048 / 1e6, tot["WRITE_SIZE"] * 1024 / 1e6, (tot["FETCH_SIZE"] * 2048 + tot["WRITE_SIZE"] * 1024) / 1e6))
PY
done
