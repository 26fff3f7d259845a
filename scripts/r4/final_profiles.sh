#!/bin/bash
# Round-4 evidence in one go (GPU box): C4 kernel stats + PMC (profile_round.sh), the headline C5 record (profile_c5.sh), the front-end
# kernels of the faithful replan chain (pmc_cmd.sh), the randomized GPU-vs-oracle parity sweep.
set -u
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r4final
bash scripts/profile_round.sh r04 > gpurun_out/r4final/profile_round.log 2>&1; tail -12 gpurun_out/r4final/profile_round.log
bash scripts/profile_c5.sh r04_c5 > gpurun_out/r4final/profile_c5.log 2>&1; tail -8 gpurun_out/r4final/profile_c5.log
bash scripts/pmc_cmd.sh r04_front "plan_kernel|decomp_kernel|jps_table|safe_path" -- python $PWD/scripts/replan_bench.py > gpurun_out/r4final/pmc_front.log 2>&1; tail -8 gpurun_out/r4final/pmc_front.log
( timeout 900 python tests/tools/parity_sweep.py 560000 600 2027 ) > gpurun_out/r4final/parity_sweep.txt 2>&1; tail -3 gpurun_out/r4final/parity_sweep.txt
