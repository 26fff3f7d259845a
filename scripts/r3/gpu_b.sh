#!/bin/bash
# round 3, call B: first run of the reduced-space solve kernel: parity tests, then the bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3b; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/pytest_parity.txt 2>&1; tail -15 $O/pytest_parity.txt
timeout 300 python bench.py --no-cpu --no-extra > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
timeout 300 python bench.py --no-cpu --no-extra --inflight 1 --steps 16 > $O/bench_solo.json 2>> $O/bench.err; tail -c 300 $O/bench_solo.json
