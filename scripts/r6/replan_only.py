"""The replan chain of bench.py alone (staged and pipelined): python scripts/r6/replan_only.py [pairs]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from faster_amd import abi
torch.cuda.init()
dev = torch.device("cuda:0")
par = abi.default_params()
out = bench.replan_leg(torch, dev, 0, par, pairs=int(sys.argv[1]) if len(sys.argv) > 1 else 65536)
print(json.dumps({k: out[k] for k in ("stages_ms", "replans_per_s", "total_ms", "pipelined")}))
