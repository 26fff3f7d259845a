"""The C5 record of bench.py alone: python scripts/r6/c5_only.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from faster_amd import abi, build as fb
torch.cuda.init()
fb.build_frontend()
out = bench.c5_leg(torch, torch.device("cuda:0"), 0, abi.default_params(), 0.05)
print(json.dumps({k: out[k] for k in ("step_ms_median", "pairs_per_s", "batches_in_flight", "safe_solved_frac")}))
