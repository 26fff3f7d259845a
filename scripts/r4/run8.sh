set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "pair or fused or c4 or glue or reference_rule or plans or c5 or forest" ) > gpurun_out/r4h/tests.txt 2>&1
tail -3 gpurun_out/r4h/tests.txt
bash scripts/r4/ab.sh r4h default default
FASTERHIP_SO=build/libfasterhip_prof.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4h/phase_pairs.txt 2>&1
grep -E "problems;|glue|hand-off|ticket|per pair" gpurun_out/r4h/phase_pairs.txt
