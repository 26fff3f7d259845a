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
# complete fh_result blocks, as bench.py gathers them: all_gather (strong scaling, one sharded batch) and gather on rank 0 (weak)
pad = np.zeros(per, dtype=abi.result_dtype); pad[: len(res)] = res
blk = torch.from_numpy(pad.view(np.uint8).reshape(-1).copy())
RES = abi.result_dtype.itemsize
g_strong = [torch.zeros(world * per * RES, dtype=torch.uint8) for _ in range(2)]
shard.gather_result_blocks(dist, blk, blk, g_strong, True, rank)
g_weak = [[torch.zeros(per * RES, dtype=torch.uint8) for _ in range(world)] for _ in range(2)] if rank == 0 else None
shard.gather_result_blocks(dist, blk, blk, g_weak, False, rank)
full = oracle.solve_batch(pr, faces, threads=2)
everything = shard.unpad_gathered(g_strong[0].numpy(), len(pr), world, abi.result_dtype)
for f in abi.result_dtype.names:   # EVERY rank holds the results of the whole batch, identical to the 1-way run
    assert np.array_equal(everything[f], full[f]), ("all_gather", f)
assert np.array_equal(g_strong[0].numpy(), g_strong[1].numpy())
if rank == 0:
    weak = shard.unpad_gathered(torch.cat(g_weak[1]).numpy(), len(pr), world, abi.result_dtype)
    for f in abi.result_dtype.names:
        assert np.array_equal(weak[f], full[f]), ("gather", f)
# the PACKED layout bench.py gathers (fh_pack_results: no dead coefficient rows, 64 + 96 N bytes per record): lossless
from faster_amd import capi
PK = capi.packed_result_size(6)
assert PK == 64 + 96 * 6
pblk = torch.from_numpy(capi.pack_results(pad, 6))
gp = [torch.zeros(world * per * PK, dtype=torch.uint8) for _ in range(2)]
shard.gather_result_blocks(dist, pblk, pblk, gp, True, rank)
unpacked = capi.unpack_results(gp[0].numpy(), world * per, 6)
everything_p = shard.unpad_gathered(unpacked.view(np.uint8), len(pr), world, abi.result_dtype)
for f in abi.result_dtype.names:
    assert np.array_equal(everything_p[f], full[f]), ("packed all_gather", f)
gpw = [[torch.zeros(per * PK, dtype=torch.uint8) for _ in range(world)] for _ in range(2)] if rank == 0 else None
shard.gather_result_blocks(dist, pblk, pblk, gpw, False, rank)
if rank == 0:
    weak_p = shard.unpad_gathered(capi.unpack_results(torch.cat(gpw[0]).numpy(), world * per, 6).view(np.uint8), len(pr), world, abi.result_dtype)
    for f in abi.result_dtype.names:
        assert np.array_equal(weak_p[f], full[f]), ("packed gather", f)
if rank == 0:
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


def _run_world(tmp_path, world):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "GLOO_OK" in outs[0][0]


def test_two_rank_gloo_sharding(tmp_path, oracle):
    _run_world(tmp_path, 2)


def test_four_rank_gloo_sharding(tmp_path, oracle):
    """37 problems over 4 ranks: shards of 10, 10, 10, 7 — the gathered fh_result blocks equal the 1-way run record for record."""
    _run_world(tmp_path, 4)


def _bench(args, env_extra=None, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=timeout)


def test_bench_launches_its_own_ranks_dry_run():
    """`python bench.py --gpus N` with no torch.distributed environment starts N ranks itself (it used to run ONE rank and
    print n_gpus 1).  --dry-run --backend gloo: the launcher, the sharding and the per-step gather without a GPU; n_gpus is the world
    size the process group really saw, one device record per rank."""
    import json

    for scaling in ("weak", "strong"):
        p = _bench(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--pairs", "37", "--scaling", scaling])
        assert p.returncode == 0, p.stderr[-2000:]
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(line) == 1, p.stdout  # ONE JSON line, from rank 0
        out = json.loads(line[0])
        assert out["n_gpus"] == 2 and out["dry_run"] and out["value"] is None and out["gather_ok"]
        assert [d["rank"] for d in out["devices"]] == [0, 1] and len({d["pid"] for d in out["devices"]}) == 2
        assert out["pairs_per_rank"] == (19 if scaling == "strong" else 37)


def test_bench_gather_modes_and_the_xgmi_budget():
    """at 21 M pairs/s a rank emits 42 GB/s of packed records, so the gather is a choice with a price.  --gather records |
    summaries | none: the dry run moves exactly those bytes over gloo, prices them per xGMI link at the assumed solve rate, and refuses a
    gather above the stated budget (a quarter of a link's one-way rate) unless told otherwise."""
    import json

    def run(extra):
        p = _bench(["--gpus", "2", "--backend", "gloo", "--dry-run", "--steps", "2", "--pairs", "37"] + extra)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        return p, (json.loads(line[0]) if len(line) == 1 else None)

    for scaling in ("weak", "strong"):
        p, out = run(["--scaling", scaling])  # default: summaries
        assert p.returncode == 0, p.stderr[-2000:]
        assert out["gather"] == "summaries" and out["gather_bytes_per_problem"] == 48 and out["gather_ok"] and out["within_budget"]
        assert abs(out["gather_GBps_per_link"] - 2 * 48 * 21.4e6 / 1e9) < 1e-9
        p, out = run(["--scaling", scaling, "--gather", "none"])
        assert p.returncode == 0 and out["gather_bytes_per_problem"] == 0 and out["gather_GBps_per_link"] == 0.0 and out["within_budget"]
        p, out = run(["--scaling", scaling, "--gather", "records"])  # 2 * 1024 B * 21.4 M/s = 43.8 GB/s per link > 19.2
        assert p.returncode != 0 and "budget" in p.stderr, (p.stdout, p.stderr[-500:])
        p, out = run(["--scaling", scaling, "--gather", "records", "--allow-over-budget"])
        assert p.returncode == 0 and out["gather_ok"] and not out["within_budget"] and out["gather_bytes_per_problem"] == 64 + 96 * 10
        assert abs(out["gather_GBps_into_busiest_gpu"] - out["gather_GBps_per_link"]) < 1e-9  # world 2: one peer
        p, out = run(["--scaling", scaling, "--gather", "records", "--assume-pairs-per-s", "1e6"])  # 2 GB/s per link: inside
        assert p.returncode == 0 and out["within_budget"]


def test_gather_traffic_arithmetic():
    from faster_amd import shard

    t = shard.gather_traffic("records", 10, 20.4e6, 8, False)
    assert abs(t["gather_GBps_per_link"] - 41.78) < 0.01 and abs(t["gather_GBps_into_busiest_gpu"] - 7 * t["gather_GBps_per_link"]) < 1e-9
    assert not t["within_budget"]
    t = shard.gather_traffic("summaries", 10, 20.4e6, 8, True)
    assert abs(t["gather_GBps_per_link"] - 1.9584) < 1e-6 and t["within_budget"] and abs(t["gather_GBps_sent_per_rank"] - 7 * 1.9584) < 1e-6
    assert shard.gather_traffic("none", 10, 20.4e6, 8, False)["gather_GBps_per_link"] == 0.0
    import torch

    rec = torch.arange(5 * 100, dtype=torch.uint8)
    h = shard.result_heads(rec, 5, 100)
    assert h.numel() == 5 * 48 and bool((h.view(5, 48) == rec.view(5, 100)[:, :48]).all())


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """--gpus 8 inside a 1-rank environment must not print a 1-GPU number; and without devices the launcher fails loudly."""
    p = _bench(["--gpus", "8", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0"})
    assert p.returncode != 0 and "refusing" in p.stderr
    import torch

    if not torch.cuda.is_available():
        p = _bench(["--gpus", "2", "--steps", "1"])
        assert p.returncode != 0 and "HIP device" in p.stderr
