#!/bin/bash
# A/B of library variants (scripts/r4/ab.sh) followed by the GPU suite on the product library:  bash scripts/r5/ab_tests.sh <tag> <variant>...
set -u
cd $GRAFT_REPO_ROOT
TAG=$1; shift
bash scripts/r4/ab.sh $TAG "$@"
bash scripts/r4/ab.sh $TAG "$@"
( time timeout 1200 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} ) > gpurun_out/$TAG/tests.txt 2>&1
grep -v ASTAR gpurun_out/$TAG/tests.txt | tail -15
