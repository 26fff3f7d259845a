#!/bin/bash
# rocprofv3 kernel stats + PMC passes (one counter set per run: gpurun refuses --pmc combined with other trace domains) of any command:
#   bash scripts/pmc_cmd.sh <tag> <kernel-name filter (regex)> -- <command ...>
# -> gpurun_out/prof_<tag>/{stats,pmc*}/ and profiles/<tag>_kernel_stats.csv, profiles/<tag>_pmc_summary.json (mean per dispatch of every
#    kernel whose name matches the filter; FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them)
set -u
TAG=$1; FILTER=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp PYTHONPATH=$R
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- "$@" > $OUT/stats.log 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -o p -- "$@" > $OUT/pmc$i.log 2>&1
  echo "pmc$i ($C): rc=$?"
done
cd $R
python - "$TAG" "$FILTER" <<'PY'
import csv, glob, collections, json, os, re, sys
tag, flt = sys.argv[1], re.compile(sys.argv[2])
root = os.getcwd()
src = os.path.join(root, "gpurun_out", "prof_" + tag)
for f in glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(root, "profiles", tag + "_kernel_stats.csv"), "w") as o:
        w = csv.DictWriter(o, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows)
    print([(r["Name"][:40], r["Calls"], r["AverageNs"]) for r in rows[:5]])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    per, names = collections.defaultdict(lambda: collections.defaultdict(float)), {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for d, c in per.items():
        k = names[d]
        if not flt.search(k): continue
        k = k.split("(")[0].replace("void ", "")
        for cn, v in c.items(): acc[k][cn].append(v)
out = {k: {c: {"mean_per_dispatch": sum(x) / len(x), "max_per_dispatch": max(x), "dispatches": len(x)} for c, x in v.items()} for k, v in acc.items()}
json.dump(out, open(os.path.join(root, "profiles", tag + "_pmc_summary.json"), "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x["mean_per_dispatch"]) for c, x in v.items() if c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")})
PY
mkdir -p $R/gpurun_out/profiles_$TAG && cp $R/profiles/${TAG}_* $R/gpurun_out/profiles_$TAG/
