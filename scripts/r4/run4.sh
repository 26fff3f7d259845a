set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q ) > gpurun_out/r4d/tests.txt 2>&1
tail -3 gpurun_out/r4d/tests.txt
bash scripts/r4/ab.sh r4d r3pb c1d1 c4d1 default c8d8 default
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4d/phase_pairs.txt 2>&1
grep -E "problems;|glue|ticket|staging|per pair|outside|result" gpurun_out/r4d/phase_pairs.txt
