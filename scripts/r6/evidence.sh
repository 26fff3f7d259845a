#!/bin/bash
# The committed evidence of round 6 on the GPU box (everything lands under gpurun_out/, the summaries are copied to profiles/ afterwards):
# rocprofv3 kernel stats + PMC passes of the C4 line and of the C5 kernel, the randomized GPU-vs-oracle sweep on the final build, the
# phase profiles of K1 and K5, the kernel trace of the replan chain's unknown-space stages.   bash scripts/r6/evidence.sh
set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6ev
git rev-parse HEAD > gpurun_out/r6ev/head.txt 2>/dev/null || true
bash scripts/profile_round.sh r06 2>&1 | grep -v ASTAR | tail -12
bash scripts/profile_c5.sh r06_c5 2>&1 | grep -v ASTAR | tail -8
( timeout 400 python tests/tools/parity_sweep.py 900000 200 6161 ) 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r6ev/parity_sweep.txt; tail -1 gpurun_out/r6ev/parity_sweep.txt
[ -f build/libfasterhip_prof.so ] && FASTERHIP_SO=$PWD/build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r6ev/phase_profile.txt
head -3 gpurun_out/r6ev/phase_profile.txt | cut -c1-200
[ -f build/libfasterhip_jpsprof.so ] && FASTERHIP_SO=$PWD/build/libfasterhip_jpsprof.so timeout 300 python scripts/jps_phase_profile.py 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r6ev/jps_phase_profile.txt
head -2 gpurun_out/r6ev/jps_phase_profile.txt | cut -c1-200
python scripts/r6/safe_chain.py 65536 7 2>&1 | tail -1 > gpurun_out/r6ev/safe_chain.json; cut -c1-120 gpurun_out/r6ev/safe_chain.json
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r6ev/safe_chain_prof -o s -- python $GRAFT_REPO_ROOT/scripts/r6/safe_chain.py 65536 7 > /dev/null 2>&1 )
( PYTHONPATH=. timeout 300 python tests/tools/decomp_ref_sweep.py 40 120 device ) 2>&1 | grep -v "ASTAR\|amdgpu.ids" > gpurun_out/r6ev/decomp_ref_sweep.txt; tail -2 gpurun_out/r6ev/decomp_ref_sweep.txt | cut -c1-250
