set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4f/phase_pairs.txt 2>&1
cat gpurun_out/r4f/phase_pairs.txt
