#!/usr/bin/env python3
"""bench.py — whole+safe trajectory solves per second on MI355X (BASELINE.json metric, config C4).

One "step" = one pass of the hot path over one batch of synthetic corridor problems:
    whole-trajectory solve (genNewTraj, N=10, <=6 polytopes)  ->  device-side whole->safe hand-off
    (R = mid sample of the whole trajectory, safe corridor = <=3 shrunk polytopes)  ->  safe-trajectory solve.
Inputs are resident in HBM before the timed region.  Each rank owns its own batch (independent problems,
weak scaling, no data-path collective); with N>1 ranks the per-pair result summaries are all-gathered over
RCCL after each step (the "batch gather" of SURVEY.md §8(e)).

Usage (driver contract):  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run (one rank per GPU).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, /opt/skills/guides/MI355X_MICROARCH.md (spec; 6.29e12 measured copy)


def algorithmic_bytes(problems, n_seg_out):
    """compulsory reads (scalars 56 B + x0/xf 144 B + 32 B per face) and writes
    (96 B per segment of coefficients + 32 B cost/dt/factor/flags + 1 B per segment assignment)."""
    nf = problems["face_off"][np.arange(len(problems)), np.clip(problems["n_poly"], 0, 8)].astype(np.int64)
    reads = 200 * len(problems) + 32 * int(nf.sum())
    writes = len(problems) * (96 * n_seg_out + 32 + n_seg_out)
    return reads + writes


def measured_traffic():
    """HBM bytes per launch of the solve kernel from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected
    in separate runs by scripts/profile_round.sh, corrected as /opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes)."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            t = json.load(open(f)).get("_hbm_traffic_per_launch_bytes")
        except Exception:
            t = None
        if t:
            best = (t["total"], os.path.basename(f))
    return best


def measured_valu_insts():
    """VALU wave-instructions per launch of solve_kernel<10> from the committed SQ_INSTS_VALU pass (profiles/r*_pmc_summary.json)."""
    import glob

    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json"))):
        try:
            v = json.load(open(f))["fh::solve_kernel<10>"]["SQ_INSTS_VALU"]["mean_per_dispatch"]
        except Exception:
            v = None
        if v:
            best = float(v)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96, help="timed steps (the end of the timed region drains the pipelines: a longer region is closer to the steady state)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=32768, help="whole+safe pairs per rank per step (C4: 32768)")
    ap.add_argument("--n-seg", type=int, default=10)
    ap.add_argument("--max-poly", type=int, default=6)
    ap.add_argument("--min-poly", type=int, default=2)
    ap.add_argument("--workload", choices=["c4", "c5"], default="c4",
                    help="c4 (default, the metric's configuration): synthetic corridors; c5: Monte-Carlo forest, corridors from the "
                         "voxel path search + ellipsoid decomposition front-end, N=15, <=8 polytopes (BASELINE config 5)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--pipeline", choices=["fused", "split"], default="split",
                    help="split: whole launch -> hand-off launch -> safe launch; fused: one launch, each wavefront takes a pair through "
                         "whole solve, hand-off and safe solve (same results)")
    ap.add_argument("--inflight", type=int, default=12,
                    help="independent pipelines (context + HIP stream + output buffers); step i runs on pipeline i %% inflight")
    args = ap.parse_args()

    # more hardware queues than the HIP default (4) so that the in-flight pipelines really run concurrently
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    from faster_amd import abi, capi, corridor, shard

    B, N = args.pairs, args.n_seg
    if args.workload == "c5":
        from faster_amd import build as fb, frontend

        fb.build_frontend()
        N = args.n_seg = 15
        args.max_poly = 8
        whole, faces, finfo = frontend.forest_batch(B, seed=5 + 1000 * rank, n_seg=N, max_poly=8)
        B = len(whole)  # pairs without a path are dropped
    else:
        whole, faces, _ = corridor.whole_batch(B, seed=3 + 1000 * rank, n_seg=N, p_choices=tuple(range(args.min_poly, args.max_poly + 1)))
    safe_t = corridor.safe_templates(whole)
    max_faces = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    d_whole, d_faces = to_dev(whole), to_dev(faces)
    RES = abi.result_dtype.itemsize
    res_words = RES // 8

    # `--inflight` independent pipelines, each with its own solver context, HIP stream and output buffers: step i runs on
    # pipeline i % inflight, so the straggler problems of one step (a single hard MIQP can take milliseconds) overlap the
    # bulk of the next step instead of idling the GPU.  Every step still does the complete work on the complete batch.
    class Pipe:
        pass

    pipes = []
    for _ in range(max(1, args.inflight)):
        pp = Pipe()
        pp.stream = torch.cuda.Stream(device=dev)
        pp.ctx = capi.Context(local_rank)
        pp.ctx.set_stream(pp.stream.cuda_stream)
        pp.d_safe = to_dev(safe_t)
        pp.d_sfaces = torch.zeros_like(d_faces)
        pp.d_wres = torch.zeros(B * RES, dtype=torch.uint8, device=dev)
        pp.d_sres = torch.zeros_like(pp.d_wres)
        pp.gather = torch.zeros((world, B, 2), dtype=torch.float64, device=dev) if world > 1 else None
        pipes.append(pp)
    torch.cuda.synchronize()
    step_no = [0]

    def step():
        pp = pipes[step_no[0] % len(pipes)]
        step_no[0] += 1
        c = pp.ctx
        if args.pipeline == "fused":
            c.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, max_faces, 0.5, 0.2, 3, pp.d_wres.data_ptr(),
                                 pp.d_safe.data_ptr(), pp.d_sfaces.data_ptr(), pp.d_sres.data_ptr())
        else:
            c.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, max_faces, pp.d_wres.data_ptr())
            c.pair_glue_device(d_whole.data_ptr(), pp.d_wres.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, pp.d_safe.data_ptr(),
                               pp.d_sfaces.data_ptr())
            c.solve_batch_device(pp.d_safe.data_ptr(), pp.d_sfaces.data_ptr(), B, N, max_faces, pp.d_sres.data_ptr())
        if world > 1:  # batch gather of the per-pair summaries (whole cost, safe cost) over RCCL/xGMI
            with torch.cuda.stream(pp.stream):
                shard.gather_step_summaries(dist, pp.d_wres, pp.d_sres, B, pp.gather.view(world * B, 2))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(len(pipes)):  # every pipeline allocates its workspace on first use: prime them all, then the W warmup steps
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    for pp in pipes:
        pp.ctx.timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    kernel_ms = np.concatenate([pp.ctx.timing_read() for pp in pipes])
    last = pipes[(step_no[0] - 1) % len(pipes)]
    wres = last.d_wres.cpu().numpy().view(abi.result_dtype)
    sres = last.d_sres.cpu().numpy().view(abi.result_dtype)
    safe_h = last.d_safe.cpu().numpy().view(abi.problem_dtype)
    sfaces_h = last.d_sfaces.cpu().numpy().view(abi.face_dtype)

    if rank == 0:
        pairs_total = world * B * args.steps
        value = pairs_total / elapsed
        # roofline of the dominant kernel (solve_kernel<10>, two launches per step: whole, safe)
        bytes_whole = algorithmic_bytes(whole, N)
        active = safe_h["n_seg"] > 0
        bytes_safe = algorithmic_bytes(safe_h[active], N) + 24 * int((~active).sum())
        bytes_per_launch = 0.5 * (bytes_whole + bytes_safe)
        avg_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9  # GB/s
        out = {
            "metric": "trajectory solves/sec (whole+safe pairs) at N=%d, deg=3" % N,
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("C4: %d whole+safe paired solves per GPU per step (N=%d segments, deg=3, <=%d polytopes whole / <=3 safe), "
                             "synthetic corridors (faster_amd/corridor.py seed 3)" % (B, N, args.max_poly)) if args.workload == "c4" else
                            ("C5: %d whole+safe paired solves per GPU per step in a random forest (20x20x3 m, 0.1 trees/m^2), corridors from "
                             "the voxel path search + ellipsoid decomposition front-end, N=15, <=8 polytopes" % B),
                "pairs_per_gpu": B,
                "pipelines_in_flight": len(pipes),
                "parallelism": "batch-sharded x%d, RCCL all_gather of result summaries" % world if world > 1 else "single GPU",
                "whole_solved_frac": float(wres["solved"].mean()),
                "safe_solved_frac": float(sres["solved"].mean()),
                "mean_bnb_nodes_whole": float(wres["nodes"].mean()),
                "mean_bnb_nodes_safe": float(sres["nodes"].mean()),
                "mean_qp_iters_per_pair": float(wres["qp_iters"].mean() + sres["qp_iters"].mean()),
                "mean_trials_per_pair": float(wres["trials"].mean() + sres["trials"].mean()),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "fh::solve_kernel<10>",
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9),
                "traffic": (measured_traffic() or (None, None))[0],
                "traffic_source": (measured_traffic() or (None, None))[1],
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_ms,
                "launches_timed": int(len(kernel_ms)),
                "pipelines_in_flight": len(pipes),
                "aggregate_achieved": (bytes_whole + bytes_safe) * args.steps / elapsed / 1e9,
                "issue_side": issue_side(measured_valu_insts(), elapsed / args.steps / 2.0) if args.workload == "c4" and N == 10 else None,
                "note": "latency/FP64-ALU bound by construction (SURVEY.md 8(d)): ~4-7 KB compulsory HBM bytes per pair; launches of "
                        "different pipelines overlap, so per-launch durations include time shared with other launches "
                        "(aggregate_achieved = all algorithmic bytes of the timed region / wall time)",
            },
        }
        if not args.no_cpu and world == 1:  # the CPU baseline is a property of the host: reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(whole, faces, safe_h, sfaces_h, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for pp in pipes:
        pp.ctx.close()


def issue_side(valu_insts, launch_s):
    """What actually bounds the kernel: wave64 VALU instructions occupy a 16-lane SIMD for 4 cycles; 256 CUs x 4 SIMDs at
    2.4 GHz.  launch_s = steady-state time per launch (two solve launches per step).  Reported next to the contract's HBM figures."""
    if not valu_insts:
        return None
    busy = valu_insts * 4.0 / (256 * 4 * 2.4e9) / launch_s
    return {"valu_wave_insts_per_launch": valu_insts, "steady_state_ms_per_launch": 1e3 * launch_s, "valu_pipes_busy_frac": busy,
            "source": "SQ_INSTS_VALU of the committed PMC pass x 4 cycles / (1024 SIMDs x 2.4 GHz)"}


def cpu_baseline(whole, faces, safe, sfaces, target_s):
    """The CPU oracle (oracle/faster_oracle.c, kind "port": Gurobi is absent) timed on a bounded sample of the SAME pairs:
    all host cores via OpenMP over problems, repeated until about `target_s` seconds of wall time have been spent.  Also
    reports the single-thread rate on a smaller sample.  Reported baseline only."""
    from oracle import oracle as orc

    orc.build()
    cores = min(os.cpu_count() or 1, orc.max_threads())

    def run(k, threads):
        t = time.perf_counter()
        orc.solve_batch(whole[:k], faces, threads=threads)
        act = safe[:k][safe[:k]["n_seg"] > 0]
        if len(act):
            orc.solve_batch(act, sfaces, threads=threads)
        return time.perf_counter() - t

    k1 = min(512, len(whole))
    t1 = run(k1, 1)                       # single thread
    k = len(whole)
    run(min(k, 4096), cores)              # warm up the thread pool
    passes, spent = 0, 0.0
    while spent < target_s and passes < 64:
        spent += run(k, cores)
        passes += 1
    return {"value": passes * k / spent, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d passes over the %d pairs of rank 0 (whole + safe solves), CPU restatement oracle/faster_oracle.c with OpenMP "
                      "over problems on %d threads; NOT Gurobi (absent)" % (passes, k, cores),
            "seconds": spent, "single_thread_value": k1 / t1, "single_thread_sample": "%d pairs" % k1}


if __name__ == "__main__":
    main()
