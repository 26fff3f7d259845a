set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
( timeout 1200 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -s ) > gpurun_out/r4j/tests.txt 2>&1
tail -25 gpurun_out/r4j/tests.txt
