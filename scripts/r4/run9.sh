set -u
cd $GRAFT_REPO_ROOT
bash scripts/r4/ab.sh r4i default default
bash scripts/profile_round.sh r04a > gpurun_out/r4i/profile_round.log 2>&1
tail -30 gpurun_out/r4i/profile_round.log
