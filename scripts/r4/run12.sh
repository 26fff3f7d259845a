set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_round4.py -m gpu -x -q -k "decomp or corridor or forest or c5 or replan or safe" ) > gpurun_out/r4l/tests.txt 2>&1
tail -4 gpurun_out/r4l/tests.txt
python - <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import torch
torch.cuda.init()
import bench
from faster_amd import abi
par = abi.default_params()
dev = torch.device("cuda", 0)
r = bench.replan_leg(torch, dev, 0, par)
print("replan_faithful stages", {k: round(v, 2) for k, v in r["stages_ms"].items()}, "total %.1f ms" % r["total_ms"])
print("mode 2 stages", {k: round(v, 2) for k, v in r["unknown_space_as_an_input"]["stages_ms"].items()})
c = bench.c5_leg(torch, dev, 0, par, 0.05)
print("c5 front end", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c["front_end"].items() if k != "path_search"}, "solver %.2f M pairs/s" % (c["pairs_per_s"] / 1e6))
PY
