set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4a/tests.txt 2>&1
tail -5 gpurun_out/r4a/tests.txt
bash scripts/r4/ab.sh r4a base tc prep default
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4a/phase_pairs.txt 2>&1
cat gpurun_out/r4a/phase_pairs.txt
