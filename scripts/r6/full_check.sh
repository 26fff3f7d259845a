#!/bin/bash
# The whole check of round 6 on the GPU box: the GPU test suite (no -x: every test runs, whatever fails), smoke(), the default bench.py line.
# The LAST gpurun of the round is this script on HEAD.   bash scripts/r6/full_check.sh [outdir]
set -u
cd $GRAFT_REPO_ROOT
out=gpurun_out/${1:-r6check}
mkdir -p $out
git rev-parse HEAD > $out/head.txt 2>/dev/null || true
( time timeout 1700 python -m pytest tests -m gpu -q -rs --durations=12 ) > $out/tests.txt 2>&1
grep -v "ASTAR\|amdgpu.ids" $out/tests.txt | tail -25
( python __graft_entry__.py --smoke 2>&1 | grep -v "ASTAR\|amdgpu.ids" | tail -3 ) | tee $out/smoke.txt
( time timeout 700 python bench.py ) > $out/bench.json 2> $out/bench.err
tail -2 $out/bench.err
python scripts/r6/bench_brief.py $out/bench.json
