set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "pair or fused or c4 or order or glue or reference_rule or plans" ) > gpurun_out/r4c/tests.txt 2>&1
tail -3 gpurun_out/r4c/tests.txt
bash scripts/r4/ab.sh r4c r3pb noahead default r3pb default
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4c/phase_pairs.txt 2>&1
grep -E "problems;|glue|ticket|staging|per pair|outside" gpurun_out/r4c/phase_pairs.txt
