#!/bin/bash
# A/B of device-library variants on the GPU box: bash scripts/r4/ab.sh <tag> <variant>...   (variant "default" = the product library)
# For every variant: bench.py (C4, timed region + the solo leg) -> one line of numbers in gpurun_out/<tag>/ab.txt
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
for v in "$@"; do
  so=$R/build/libfasterhip_$v.so
  [ "$v" = default ] && so=$R/faster_amd/libfasterhip.so
  FASTERHIP_SO=$so timeout 300 python bench.py --no-cpu --solo-only --steps ${STEPS:-64} --warmup 8 ${BENCH_ARGS:-} > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - "$v" $OUT/bench_$v.json <<'PY' | tee -a $OUT/ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    c, r = d["config"], d["roofline"]
    print("%-10s %6.2f M pairs/s  %.3f ms/step | solo %.3f ms (%.2f M/s) | iters/pair %.2f nodes %.2f/%.2f | solved %.4f/%.4f | share %s" % (
        sys.argv[1], d["value"] / 1e6, d["ms_per_step"], r["solo"]["step_ms_median"], r["solo"]["pairs_per_s"] / 1e6, c["mean_qp_iters_per_pair"],
        c["mean_bnb_nodes_whole"], c["mean_bnb_nodes_safe"], c["whole_solved_frac"], c["safe_solved_frac"], c["share_stats_last_launch"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
