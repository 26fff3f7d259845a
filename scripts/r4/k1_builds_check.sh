#!/bin/bash
# the two builds of the solve kernel: bit-equality tests + bench.py (its roofline.solo_two_wavefronts_per_simd record) (GPU box)
cd "$(dirname "$0")/../.."
export PYTHONPATH=$PWD
timeout 900 python -m pytest tests -x -q -m gpu -k "two_builds or launch_order_or_publishing" 2>&1 | tail -3
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_two_builds.json
python -c "
import json
d=json.loads(open('gpurun_out/bench_two_builds.json').read()); r=d['roofline']
print('value %.2f M; solo %.3f ms frac %.2e; two-wave solo %.3f ms (%.2f M/s), same results: %s' % (d['value']/1e6, r['solo']['step_ms_median'], r['frac'], r['solo_two_wavefronts_per_simd']['step_ms_median'], r['solo_two_wavefronts_per_simd']['pairs_per_s']/1e6, r['solo_two_wavefronts_per_simd']['same_results_as_the_throughput_build']))"
