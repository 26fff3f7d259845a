#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/r2j
timeout 600 python -m pytest tests/test_gurobi_semantics.py -m gpu -x -q 2>&1 | tail -5
timeout 600 python scripts/gurobi_semantics_report.py r02 2>&1 | tail -8
cp profiles/r02_gurobi_semantics.json gpurun_out/r2j/
