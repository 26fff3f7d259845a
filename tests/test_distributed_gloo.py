"""N>1 path on CPU: world_size-2 gloo run of the batch sharding + summary gather used by bench.py (faster_amd/shard.py).
The solver here is the CPU oracle (allowed in tests): the point is the partition/gather logic, not the kernels."""
import os
import socket
import subprocess
import sys

import numpy as np

from faster_amd import corridor, shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from faster_amd import corridor, shard
from oracle import oracle
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
pr, faces, _ = corridor.whole_batch(37, seed=4, n_seg=6, p_choices=(1, 2, 3))     # odd size: uneven shards
lo, hi = shard.shard_range(len(pr), rank, world)
lpr, lfaces = shard.shard_batch(pr, faces, rank, world)
res = oracle.solve_batch(lpr, lfaces, threads=2)
per = -(-len(pr) // world)
g = shard.all_gather_rows(dist, torch.from_numpy(shard.summaries(res)), per)
dist.barrier()
# the per-step gather of bench.py (same function, CPU tensors + gloo here, device tensors + RCCL on the GPUs)
from faster_amd import abi
wt = torch.from_numpy(res.view(np.uint8).reshape(-1).copy())
out = torch.zeros((world * len(res), 2), dtype=torch.float64)
if len(set(dist_sizes := [shard.shard_range(len(pr), r, world)[1] - shard.shard_range(len(pr), r, world)[0] for r in range(world)])) == 1:
    shard.gather_step_summaries(dist, wt, wt, len(res), out)
else:  # uneven shards: pad to the common size as bench.py's equal per-rank batches never need to
    pad = np.zeros(per, dtype=abi.result_dtype); pad[: len(res)] = res
    wt = torch.from_numpy(pad.view(np.uint8).reshape(-1).copy())
    out = torch.zeros((world * per, 2), dtype=torch.float64)
    shard.gather_step_summaries(dist, wt, wt, per, out)
    mine = out[rank * per: rank * per + len(res)].numpy()
    assert np.array_equal(mine[:, 0], res["cost"]) and np.array_equal(mine[:, 1], res["cost"])
if rank == 0:
    full = oracle.solve_batch(pr, faces, threads=2)
    got = np.concatenate([g[r * per: r * per + (shard.shard_range(len(pr), r, world)[1] - shard.shard_range(len(pr), r, world)[0])].numpy() for r in range(world)])
    assert got.shape[0] == len(pr)
    assert np.array_equal(got, shard.summaries(full)), "sharded results differ from the single-process run"
    print("GLOO_OK", int(full["solved"].sum()))
dist.destroy_process_group()
'''


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 32768):
        for world in (1, 2, 4, 8):
            spans = [shard.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]


def test_shard_batch_rebases_faces():
    pr, faces, _ = corridor.whole_batch(10, seed=9)
    a, fa = shard.shard_batch(pr, faces, 1, 3)
    lo, hi = shard.shard_range(10, 1, 3)
    assert len(a) == hi - lo and a["face_begin"][0] == 0
    for i in range(len(a)):
        n = a["face_off"][i][a["n_poly"][i]]
        g0 = pr["face_begin"][lo + i]
        assert np.array_equal(fa["b"][a["face_begin"][i]: a["face_begin"][i] + n], faces["b"][g0: g0 + n])


def test_two_rank_gloo_sharding(tmp_path, oracle):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "GLOO_OK" in outs[0][0]
