"""bench.py's `replan_faithful` leg alone (the device pipeline of one Faster::replan per pair, 65536 pairs): the command that
scripts/pmc_cmd.sh profiles for the front-end kernels of that pipeline (plan_kernel, decomp_kernel with the unknown voxels in)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from faster_amd import abi  # noqa: E402

torch.cuda.init()
r = bench.replan_leg(torch, torch.device("cuda", 0), 0, abi.default_params(), pairs=int(sys.argv[1]) if len(sys.argv) > 1 else 65536, reps=2)
print(json.dumps({k: v for k, v in r.items() if k in ("stages_ms", "total_ms", "replans_per_s", "unknown_space_as_an_input")}))
