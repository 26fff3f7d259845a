#!/bin/bash
# The whole check of a round on the GPU box: the GPU test suite (no -x: every test runs, whatever fails), smoke(), the default
# bench.py line.  The LAST gpurun of a round is this script on HEAD.   bash scripts/r5/full_check.sh [outdir]
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r5check}
mkdir -p $out
git rev-parse HEAD > $out/head.txt 2>/dev/null || true
( time timeout 1500 python -m pytest tests -m gpu -q -rs --durations=12 ) > $out/tests.txt 2>&1
grep -v ASTAR $out/tests.txt | tail -25
( python __graft_entry__.py --smoke 2>&1 | grep -v ASTAR | tail -3 ) | tee $out/smoke.txt
( time timeout 600 python bench.py ) > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err
python - $out/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value %.2f M pairs/s %.3f ms/step; solo %.3f ms; kernel %s traffic %s (%s) wasted_x %s" % (d["value"] / 1e6, d["ms_per_step"], r["solo"]["step_ms_median"], r["kernel"], r["traffic"], r["traffic_source"], r.get("wasted_x")))
print("c5 %.2f M; e2e %.2f M; replan %.1f ms %s; latency %.3f ms; cpu %.1f k (%s iters/pair vs gpu %.1f)" % (d["c5"]["pairs_per_s"] / 1e6, d["e2e_with_copies"]["pairs_per_s"] / 1e6, d["replan_faithful"]["total_ms"], d["replan_faithful"]["stages_ms"], d["single_replan_latency_ms"]["median_ms"], d["cpu_baseline"]["value"] / 1e3, d["cpu_baseline"].get("mean_qp_iters_per_pair"), d["config"]["mean_qp_iters_per_pair"]))
PY
