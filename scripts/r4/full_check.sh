#!/bin/bash
# The whole check of a round on the GPU box: the GPU test suite, smoke(), the full bench.py, the C5 workload, the cell-record
# comparison of the path search and its phase profile (needs build/libfasterhip_jpsprof.so: bash scripts/build_variant.sh jpsprof -DFHP_PROFILE).
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4check
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4check/tests.txt 2>&1
tail -4 gpurun_out/r4check/tests.txt
python __graft_entry__.py --smoke 2>&1 | tail -3
( time timeout 600 python bench.py ) > gpurun_out/r4check/bench_full.json 2> gpurun_out/r4check/bench_full.err
tail -2 gpurun_out/r4check/bench_full.err
python -c "
import json
d=json.loads(open('gpurun_out/r4check/bench_full.json').read().strip().splitlines()[-1])
print('value %.2f M pairs/s %.3f ms/step; solo %.3f ms; c5 %.2f M; e2e %.2f M; replan %.1f ms; latency %.3f ms; cpu %.1f k' % (d['value']/1e6, d['ms_per_step'], d['roofline']['solo']['step_ms_median'], d['c5']['pairs_per_s']/1e6, d['e2e_with_copies']['pairs_per_s']/1e6, d['replan_faithful']['total_ms'], d['single_replan_latency_ms']['median_ms'], d['cpu_baseline']['value']/1e3))
"
timeout 300 python bench.py --workload c5 --pairs 65536 --no-cpu --no-extra --steps 16 --warmup 2 > gpurun_out/r4check/bench_c5.json 2> gpurun_out/r4check/bench_c5.err; tail -c 600 gpurun_out/r4check/bench_c5.json | head -c 300; python -c "
import json
d=json.loads(open('gpurun_out/r4check/bench_c5.json').read().strip().splitlines()[-1]); print('\nc5 workload (8 in flight): %.2f M pairs/s, %.2f ms/step, safe solved %.3f' % (d['value']/1e6, d['ms_per_step'], d['config']['safe_solved_frac']))"

timeout 900 python scripts/records_bench.py 65536 gpurun_out/r4check/records.json 2>&1 | grep -v ASTAR | cut -c1-230
FASTERHIP_SO=build/libfasterhip_jpsprof.so timeout 600 python scripts/jps_phase_profile.py 65536 0 2>&1 | grep -v ASTAR | tee gpurun_out/r4check/jps_phases.txt
