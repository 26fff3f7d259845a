"""The safe corridor of a replan on the device against the host decomposition of the explicit cloud [unknown voxels | occupied points],
row for row and bit for bit, on forests of several seeds (the body of tests/test_gpu_round3.py::test_safe_corridor_decomposed_around_r_on_the_device;
its lists hold 2-3 k points: K4's hybrid list, head in LDS and tail in the workspace).   PYTHONPATH=. python tests/tools/safe_corridor_sweep.py [seeds] [pairs checked per seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # (torch before the HIP library: one HIP runtime in the process)
torch.cuda.init()
from faster_amd import build as fb, capi
from oracle import oracle
import test_gpu_round3 as T

fb.build_frontend()
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_check = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ctx = capi.Context(0)
t0 = time.time()
done = 0
for seed in range(100, 100 + seeds):
    for r_known in (4.0, 0.2):
        T.test_safe_corridor_decomposed_around_r_on_the_device(ctx, oracle, r_known, seed=seed, n_check=n_check)
        done += 1
        print("seed %d r_known %.1f: safe paths and polytopes equal the host's (bit for bit) | %d s" % (seed, r_known, time.time() - t0), flush=True)
print("SAFE CORRIDOR SWEEP DONE: %d scenes, every checked pair equal" % done)
ctx.close()
