#!/bin/bash
# PMC passes for the front-end kernels (plan_kernel<true>, decomp_kernel): the same counter sets as scripts/profile_round.sh, one pass each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_r03_front_pmc
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- python $R/scripts/front_bench.py 65536 0 > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($C): rc=$?"
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "plan_kernel" in k or "decomp_kernel" in k or "jps_table" in k:
            acc[k.split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {k: {c: {"mean_per_dispatch": sum(x) / len(x), "dispatches": len(x)} for c, x in v.items()} for k, v in acc.items()}
json.dump(out, open("$OUT/r03_front_pmc_summary.json", "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x["mean_per_dispatch"]) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")})
PY
