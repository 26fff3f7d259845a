"""Batch sharding across the GPUs of one node.

Every genNewTraj() is independent, and a whole+safe pair never leaves its GPU, so the only communication is the
gather of per-pair result summaries (RCCL all_gather over xGMI when the backend is "nccl"; gloo in CPU tests).
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous block partition: rank r owns [lo, hi) with ceil(n/world) items per rank (pairs are never split)."""
    per = -(-n // world)
    lo = min(rank * per, n)
    hi = min(lo + per, n)
    return lo, hi


def shard_batch(problems, faces, rank, world):
    """Slice a (problems, faces) batch for one rank, rebasing face_begin onto the rank-local face array."""
    lo, hi = shard_range(len(problems), rank, world)
    pr = problems[lo:hi].copy()
    if hi == lo:
        return pr, faces[:0].copy()
    nf = pr["face_off"][np.arange(hi - lo), np.clip(pr["n_poly"], 0, pr["face_off"].shape[1] - 1)]
    f_lo = int(pr["face_begin"].min())
    f_hi = int((pr["face_begin"] + nf).max())
    pr["face_begin"] -= f_lo
    return pr, faces[f_lo:f_hi].copy()


def summaries(results):
    """[n, 4] float64: solved, factor, dt, cost — what rank 0 needs from every pair."""
    out = np.zeros((len(results), 4), dtype=np.float64)
    out[:, 0] = results["solved"]
    out[:, 1] = results["factor"]
    out[:, 2] = results["dt"]
    out[:, 3] = results["cost"]
    return out


def all_gather_rows(dist, local, per_rank):
    """all_gather of a [<=per_rank, k] tensor padded to per_rank rows; returns [world*per_rank, k]."""
    import torch

    world = dist.get_world_size()
    pad = torch.zeros((per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = torch.zeros((world * per_rank, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad)
    return out


# ---- what the per-step gather moves, and what that costs on xGMI (SURVEY.md 8(e) priced the gather at 10 k solves/s) ----
# One xGMI link of an MI355X: ~153.6 GB/s both directions together, i.e. ~76.8 GB/s one way, point to point between two GPUs (7 links
# per GPU).  In the weak-scaling gather every rank sends to rank 0 over ITS link to rank 0; in the all_gather of a sharded batch every
# rank sends its block to each of the other ranks over the link to that rank.  The stated budget: a gather may use a quarter of a
# link's one-way rate — beyond that the 8-GPU number would measure the gather, not the solver.
XGMI_LINK_ONE_WAY_GBPS = 76.8
XGMI_GATHER_BUDGET_FRACTION = 0.25
HEAD_BYTES = 48   # the head of an fh_result: solved, trials, status, nodes, qp_iters, kflops, factor, dt, cost
GATHER_MODES = ("records", "summaries", "none")


def gather_bytes_per_problem(mode, n_seg):
    """records: the packed fh_result (64 + 96 N bytes, fh_pack_results); summaries: its 48-byte head (flags, counters, factor, dt, cost);
    none: results are consumed where they are produced."""
    if mode == "records":
        return 64 + 96 * int(n_seg)
    if mode == "summaries":
        return HEAD_BYTES
    if mode == "none":
        return 0
    raise ValueError(mode)


def gather_traffic(mode, n_seg, pairs_per_s_per_rank, world, strong):
    """GB/s the per-step gather puts on xGMI at a given solve rate (two problems per pair): what one rank sends, what one link carries,
    what the busiest GPU receives — against the stated budget per link."""
    per_rank = 2.0 * gather_bytes_per_problem(mode, n_seg) * float(pairs_per_s_per_rank) / 1e9
    peers = max(int(world) - 1, 0)
    budget = XGMI_LINK_ONE_WAY_GBPS * XGMI_GATHER_BUDGET_FRACTION
    return {"mode": mode, "bytes_per_pair": 2 * gather_bytes_per_problem(mode, n_seg), "pairs_per_s_per_rank": float(pairs_per_s_per_rank),
            "gather_GBps_sent_per_rank": per_rank * (peers if strong else (1 if peers else 0)),
            "gather_GBps_per_link": per_rank if peers else 0.0,
            "gather_GBps_into_busiest_gpu": per_rank * peers,
            "link_one_way_GBps": XGMI_LINK_ONE_WAY_GBPS, "budget_GBps_per_link": budget, "within_budget": bool(per_rank <= budget)}


def result_heads(results_u8, n, record_bytes):
    """[n * record_bytes] uint8 tensor of fh_result (or packed) records -> [n * HEAD_BYTES]: the 48-byte heads, contiguous (a strided copy on
    the current stream: plumbing, the solver is not involved)."""
    return results_u8.view(n, record_bytes)[:, :HEAD_BYTES].contiguous().view(-1)


def gather_result_blocks(dist, whole_results, safe_results, gather, strong, rank):
    """The per-step "batch gather" of bench.py: complete `fh_result` blocks, not summaries.
    whole_results / safe_results: uint8 tensors of this rank's per_rank records (a shorter shard is padded to per_rank).
    strong (one batch sharded over the ranks): all_gather, gather = [whole_out, safe_out], each world*per_rank records — every
    rank ends the step with the results of the whole batch (row r*per_rank + i = problem shard_range(r)[0] + i).
    weak (a batch per rank): gather on rank 0, gather = [[per-rank buffers], [per-rank buffers]] there, None elsewhere.
    Device tensors with the nccl backend = RCCL over xGMI; CPU tensors with gloo in the tests.  Runs on the current stream."""
    if strong:
        dist.all_gather_into_tensor(gather[0], whole_results)
        dist.all_gather_into_tensor(gather[1], safe_results)
    else:
        dist.gather(whole_results, gather_list=gather[0] if rank == 0 else None, dst=0)
        dist.gather(safe_results, gather_list=gather[1] if rank == 0 else None, dst=0)
    return gather


def unpad_gathered(blocks, n, world, dtype):
    """[world*per_rank] records gathered from contiguous shards -> the n records of the original batch, in order."""
    import numpy as np

    per = -(-n // world)
    rec = np.asarray(blocks).view(dtype)
    return np.concatenate([rec[r * per: r * per + (shard_range(n, r, world)[1] - shard_range(n, r, world)[0])] for r in range(world)])


def gather_step_summaries(dist, whole_results, safe_results, n, out):
    """The per-step "batch gather" of bench.py: every rank contributes (whole cost, safe cost) of its n pairs; `out` is
    [world*n, 2].  whole_results / safe_results are uint8 tensors holding n `fh_result` records (device tensors with the
    nccl backend = RCCL over xGMI; CPU tensors with gloo in the tests).  Runs on the current stream."""
    import torch

    from . import abi

    words = abi.result_dtype.itemsize // 8
    cost_word = abi.result_dtype.fields["cost"][1] // 8
    ww = whole_results.view(torch.float64).view(n, words)[:, cost_word]
    sw = safe_results.view(torch.float64).view(n, words)[:, cost_word]
    dist.all_gather_into_tensor(out, torch.stack([ww, sw], dim=1))
    return out
