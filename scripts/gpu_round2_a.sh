#!/bin/bash
# first GPU pass of round 2: parity tests, work-sharing diagnostic, quick bench legs
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -15 $O/pytest.log
timeout 600 python scripts/share_diag.py > $O/share_diag.log 2>&1; echo "diag rc=$?"
cat $O/share_diag.log | tail -40
timeout 600 python bench.py --no-cpu --inflight 1 --steps 24 --warmup 4 > $O/bench_if1.json 2> $O/bench_if1.err; echo "bench if1 rc=$?"; cat $O/bench_if1.json | cut -c1-400
timeout 600 python bench.py --no-cpu --inflight 1 --steps 24 --warmup 4 --pipeline fused > $O/bench_if1_fused.json 2> $O/bench_if1_fused.err; echo "bench if1 fused rc=$?"; cat $O/bench_if1_fused.json | cut -c1-400
timeout 600 python bench.py --no-cpu > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; cat $O/bench_default.json | cut -c1-400
