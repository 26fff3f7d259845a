set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4k/tests.txt 2>&1
tail -6 gpurun_out/r4k/tests.txt
( time timeout 600 python bench.py ) > gpurun_out/r4k/bench_full.json 2> gpurun_out/r4k/bench_full.err
tail -3 gpurun_out/r4k/bench_full.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4k/bench_full.json").read().strip().splitlines()[-1])
print("value %.2f M pairs/s, %.3f ms/step" % (d["value"] / 1e6, d["ms_per_step"]))
for k in ("roofline", "e2e_with_copies", "c5", "replan_faithful", "single_replan_latency_ms", "cpu_baseline"):
    v = d.get(k)
    if isinstance(v, dict):
        print(k, json.dumps({a: b for a, b in v.items() if not isinstance(b, str) or len(b) < 80})[:1500])
print({k: v for k, v in d["config"].items() if "literal" in k})
PY
