#!/bin/bash
# bash scripts/r6_prof.sh <tag> : phase profile of the FH_PROFILE build (build/libfasterhip_prof.so) + the A/B line of the product library
set -u
cd $GRAFT_REPO_ROOT
TAG=${1:-r6prof}
mkdir -p gpurun_out/$TAG
FASTERHIP_SO=$PWD/build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 2>&1 | grep -v amdgpu.ids > gpurun_out/$TAG/phase_profile.txt
cat gpurun_out/$TAG/phase_profile.txt | cut -c1-400
STEPS=96 bash scripts/r4/ab.sh $TAG default
