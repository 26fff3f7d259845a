#!/bin/bash
# The committed evidence of round 5 on the GPU box (everything lands under gpurun_out/, the summaries are copied to profiles/ afterwards):
# rocprofv3 kernel stats + PMC passes of the C4 line and of the C5 kernel, the randomized GPU-vs-oracle sweep on the final build, the
# decomposition reference sweep with K4 beside the host restatement, the phase profile.   bash scripts/r5/evidence.sh
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ev
bash scripts/profile_round.sh r05 2>&1 | grep -v ASTAR | tail -12
bash scripts/profile_c5.sh r05_c5 2>&1 | grep -v ASTAR | tail -8
( timeout 400 python tests/tools/parity_sweep.py 900000 200 5151 ) 2>&1 | grep -v ASTAR > gpurun_out/r5ev/parity_sweep.txt; tail -1 gpurun_out/r5ev/parity_sweep.txt
( PYTHONPATH=. timeout 300 python tests/tools/decomp_ref_sweep.py 40 120 device ) 2>&1 | grep -v ASTAR > gpurun_out/r5ev/decomp_ref_sweep.txt; tail -2 gpurun_out/r5ev/decomp_ref_sweep.txt
[ -f build/libfasterhip_prof.so ] && FASTERHIP_SO=build/libfasterhip_prof.so python scripts/phase_profile.py 32768 pairs 2>&1 | grep -v ASTAR > gpurun_out/r5ev/phase_profile.txt
