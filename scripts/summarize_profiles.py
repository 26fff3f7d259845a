"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "prof_" + tag)
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

# 1. kernel stats (rocprofv3 --kernel-trace --stats): the driver's configuration, and one batch at a time ("solo")
for sub, suffix in (("stats", "_kernel_stats.csv"), ("stats_solo", "_kernel_stats_solo.csv")):
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        with open(os.path.join(dst, tag + suffix), "w") as o:
            w = csv.DictWriter(o, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
        print(sub, "kernel stats:", [(r["Name"][:44], r["Calls"], r["AverageNs"]) for r in rows[:4]])

# 2. PMC counters, summed over the dimension instances, averaged per dispatch of each kernel
summary = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(src, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for d, c in per.items():
        k = names[d]
        if "fh::" not in k:
            continue
        k = k.split("(")[0].replace("void ", "")
        for cn, v in c.items():
            summary[k][cn].append(v)
out = {}
for k, c in summary.items():
    out[k] = {cn: {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)} for cn, v in c.items()}
sk = next((k for k in out if "solve_kernel" in k), None)
if sk and "FETCH_SIZE" in out[sk] and "WRITE_SIZE" in out[sk]:
    # /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of
    # the bytes of a wide coalesced read => doubled.  WRITE_SIZE is uncalibrated (taken as is).
    fetch = out[sk]["FETCH_SIZE"]["mean_per_dispatch"] * 1024 * 2
    write = out[sk]["WRITE_SIZE"]["mean_per_dispatch"] * 1024
    out["_kernel"] = sk
    out["_hbm_traffic_per_launch_bytes"] = {"kernel": sk, "fetch_corrected": fetch, "write": write, "total": fetch + write,
                                            "note": "FETCH_SIZE x2 (gfx950 correction), WRITE_SIZE as reported; KiB units"}
json.dump(out, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out.get("_hbm_traffic_per_launch_bytes", {}), indent=1))
