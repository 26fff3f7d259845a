set -u
cd $GRAFT_REPO_ROOT
bash scripts/r4/ab.sh r4g default nt default nt
