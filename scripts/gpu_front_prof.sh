#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export PYTHONUNBUFFERED=1 PYTHONPATH=$R TMPDIR=/tmp
O=$R/gpurun_out/front; mkdir -p $O
timeout 600 python scripts/front_bench.py 65536 8192 $O/r02_frontend.json 2>&1 | grep -v amdgpu | tail -2
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python $R/scripts/front_bench.py 65536 0 > $O/stats.log 2>&1
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-200; cp $f $O/r02_frontend_kernel_stats.csv
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc$i -o p -- python $R/scripts/front_bench.py 65536 0 > $O/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, json, collections
O="gpurun_out/front"
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+"/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k=row["Kernel_Name"]
        name = "plan_kernel" if "plan_kernel" in k else "decomp_kernel" if "decomp_kernel" in k else None
        if name: acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
out={}
for k,v in acc.items():
    out[k]={c:{"max_per_dispatch":max(x),"dispatches":len(x)} for c,x in v.items()}
json.dump(out,open(O+"/r02_frontend_pmc.json","w"),indent=1)
pk=out.get("plan_kernel",{})
if "FETCH_SIZE" in pk and "WRITE_SIZE" in pk:
    print("plan_kernel (largest dispatch = the 65536-query launch): fetch x2 KiB -> GB %.2f, write GB %.2f" % (pk["FETCH_SIZE"]["max_per_dispatch"]*2*1024/1e9, pk["WRITE_SIZE"]["max_per_dispatch"]*1024/1e9))
print({c: v["max_per_dispatch"] for c, v in pk.items()})
PY
