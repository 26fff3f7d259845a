set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4e
BENCH_ARGS="--wg-per-cu 8" bash scripts/r4/ab.sh r4e default w2 default w2
FASTERHIP_SO=build/libfasterhip_profic.so timeout 300 python scripts/phase_profile.py 32768 pairs > gpurun_out/r4e/phase_pairs.txt 2>&1
grep -E "problems;|dt_initial|per pair" gpurun_out/r4e/phase_pairs.txt
