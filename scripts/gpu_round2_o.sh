#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
PAIR_MARGIN=0.05 timeout 300 python -u scripts/share_diag.py 32768 2>&1 | grep -E "fused|==" | tail -5
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for nf in 1 8 12; do timeout 300 python bench.py --no-cpu --no-extra --inflight $nf --steps 48 > /tmp/b.json 2>/tmp/b.err; python - <<PY
import json
d=json.load(open("/tmp/b.json")); print("fused inflight $nf: %.2f M pairs/s, %.2f ms/step" % (d["value"]/1e6, d["ms_per_step"]))
PY
done
O=$R/gpurun_out/prof_x; rm -rf $O; mkdir -p $O; cd /tmp
for pl in fused; do for C in FETCH_SIZE WRITE_SIZE; do
timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${pl}_$C -o p -- python $R/bench.py --no-cpu --no-extra --pipeline $pl --inflight 1 --steps 8 --warmup 2 > $O/${pl}_$C.log 2>&1
python - <<PY
import csv, glob, collections
per=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob("$O/${pl}_$C/**/*counter_collection.csv", recursive=True):
    d=collections.defaultdict(float); names={}
    for r in csv.DictReader(open(f)):
        d[r["Dispatch_Id"]]+=float(r["Counter_Value"]); names[r["Dispatch_Id"]]=r["Kernel_Name"]
    for k,v in d.items():
        if "solve_kernel" in names[k]: per[names[k][:40]]+=v; n[names[k][:40]]+=1
for k in per: print("$pl $C", k, "mean per dispatch %.0f KiB over %d dispatches" % (per[k]/n[k], n[k]))
PY
done; done
