#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3pc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit cycles --pc-sampling-method stochastic --pc-sampling-interval 1048576 --output-format csv -d $O/pc -o pc -- python $R/bench.py --no-cpu --no-extra --inflight 1 --steps 8 --warmup 2 > $O/pc.log 2>&1
echo rc=$?; tail -5 $O/pc.log | cut -c1-300; ls -la $O/pc | head; 
