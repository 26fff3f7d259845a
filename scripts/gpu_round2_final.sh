#!/bin/bash
# Round-2 measurement pass on the GPU box: tests, the driver's bench line, comparison legs, rocprofv3 evidence, C5, parity sweep.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=$R/gpurun_out/r2final; mkdir -p $O
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee $O/pytest.txt
echo "== bench (driver default)"; timeout 900 python bench.py > $O/r02_bench.json 2> $O/bench.err; echo rc=$?
run() { name=$1; shift; timeout 600 python bench.py --no-cpu --no-extra "$@" > $O/$name.json 2> $O/$name.err; python - <<PY
import json
try:
    d=json.load(open("$O/$name.json")); print("$name: %.3f M pairs/s, %.2f ms/step, avg launch %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["avg_launch_ms"]), "safe solved %.3f" % d["config"]["safe_solved_frac"])
except Exception as e: print("$name failed", e)
PY
}
run fused_if1 --inflight 1
run fused_if2 --inflight 2
run fused_if4 --inflight 4
run fused_if12 --inflight 12
run split_if12 --pipeline split --inflight 12
run noshare_if1 --no-share --inflight 1
run noshare_if8 --no-share --inflight 8
run literal_if8 --r-margin -1
run literal_split12 --r-margin -1 --pipeline split --inflight 12
run literal_if1 --r-margin -1 --inflight 1
run c5_65536 --workload c5 --pairs 65536 --steps 8 --warmup 2
run c5_host_front --workload c5 --front host --pairs 65536 --steps 8 --warmup 2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
echo "== profiles"; bash scripts/profile_round.sh r02 2>&1 | tail -8
echo "== parity sweep"; timeout 400 python tests/tools/parity_sweep.py 150000 240 2>&1 | tail -3 | tee $O/r02_parity_sweep.txt
cp $O/r02_parity_sweep.txt $R/gpurun_out/profiles_r02/ 2>/dev/null
cp $O/r02_bench.json $R/gpurun_out/profiles_r02/ 2>/dev/null
