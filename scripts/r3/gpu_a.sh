#!/bin/bash
# round 3, call A: baseline numbers before the reduced-space rewrite (phase profile, forced 3 waves/SIMD build)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3a; mkdir -p $O
cd $R
python bench.py --no-cpu --no-extra > $O/bench_base.json 2> $O/bench_base.err; tail -c 600 $O/bench_base.json
python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/bench_base_solo.json 2>> $O/bench_base.err; tail -c 300 $O/bench_base_solo.json
FASTERHIP_SO=$R/build/variants/libfh_lb3.so python bench.py --no-cpu --no-extra > $O/bench_lb3.json 2> $O/bench_lb3.err; tail -c 300 $O/bench_lb3.json
FASTERHIP_SO=$R/build/variants/libfh_prof.so python scripts/phase_profile.py 8192 > $O/phase.txt 2>&1; cat $O/phase.txt
