set -u
cd $GRAFT_REPO_ROOT
bash scripts/r4/ab.sh r4m default w2
BENCH_ARGS="--wg-per-cu 10" bash scripts/r4/ab.sh r4m default
BENCH_ARGS="--wg-per-cu 9" bash scripts/r4/ab.sh r4m default
BENCH_ARGS="--inflight 4" bash scripts/r4/ab.sh r4m default w2
BENCH_ARGS="--inflight 12" bash scripts/r4/ab.sh r4m default
