set -u
cd $GRAFT_REPO_ROOT
bash scripts/r4/ab.sh r4b r3pb base prep r3pb base
FASTERHIP_SO=build/libfasterhip_profbase.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4b/phase_pairs_base.txt 2>&1
cat gpurun_out/r4b/phase_pairs_base.txt
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4b/phase_pairs.txt 2>&1
cat gpurun_out/r4b/phase_pairs.txt
