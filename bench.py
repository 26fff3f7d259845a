#!/usr/bin/env python3
"""bench.py — whole+safe trajectory solves per second on MI355X (BASELINE.json metric, config C4).

One "step" = one pass of the hot path over one batch of synthetic corridor problems:
    whole-trajectory solve (genNewTraj, N=10, <=6 polytopes)  ->  device-side whole->safe hand-off
    (R = mid sample of the whole trajectory, safe corridor = <=3 polytopes around R)  ->  safe-trajectory solve.
Inputs are resident in HBM before the timed region.

Multi-GPU (one process per GPU, torch.distributed over RCCL):
  --scaling weak   (default) every rank owns its own batch of --pairs pairs; after each step the complete fh_result blocks of
                   all ranks are gathered on rank 0 (RCCL send/recv over xGMI);
  --scaling strong ONE batch of --pairs pairs (BASELINE config 4: 32768) is sharded over the ranks in contiguous blocks
                   (faster_amd/shard.py) and the complete fh_result blocks are all-gathered, so every rank ends a step with
                   the results of the whole batch.
There is no data-path collective inside a step: problems are independent and a pair never leaves its GPU.

Usage (driver contract):  python bench.py --gpus N --steps K --warmup W
N>1 is launched by torch.distributed.run (one rank per GPU).  Prints ONE JSON line on rank 0.

Besides the timed region (K steps, `--inflight` independent pipelines) the N=1 run measures, outside the timed region:
a single batch alone on the GPU (`roofline.solo`: no overlap between launches; `roofline.solo_two_wavefronts_per_simd`: the same with
the kernel build a caller who waits for one batch selects, fh_sched.workgroups_per_cu <= 8), the PCIe-inclusive host-pointer path
(`e2e_with_copies`), the FP64 flop rate against the measured FP64 FMA peak (`roofline.compute`) and the CPU baseline.
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
FP64_VECTOR_SPEC_TFLOPS = 78.6  # AMD datasheet; the measured FMA peak is what `roofline.compute` divides by


def algorithmic_bytes(problems, n_seg_out):
    """compulsory reads (scalars 56 B + x0/xf 144 B + 32 B per face) and writes
    (96 B per segment of coefficients + 32 B cost/dt/factor/flags + 1 B per segment assignment)."""
    nf = problems["face_off"][np.arange(len(problems)), np.clip(problems["n_poly"], 0, 8)].astype(np.int64)
    reads = 200 * len(problems) + 32 * int(nf.sum())
    writes = len(problems) * (96 * n_seg_out + 32 + n_seg_out)
    return reads + writes


ROUND = "r06"   # the round whose profiles/ this bench line may quote (never an older round's counters for a newer kernel)


def measured_traffic(kernel_name):
    """HBM bytes per launch of the solve kernel from THIS round's committed rocprofv3 PMC passes (profiles/<ROUND>_pmc_summary.json:
    FETCH_SIZE / WRITE_SIZE collected in separate runs by scripts/profile_round.sh, corrected as /opt/skills/guides/MI355X_MICROARCH.md
    §HBM prescribes) — only if that file is about exactly the kernel instantiation that ran (fh_last_launch); otherwise None."""
    f = os.path.join(ROOT, "profiles", ROUND + "_pmc_summary.json")
    try:
        d = json.load(open(f))
        t = d.get("_hbm_traffic_per_launch_bytes")
        k = d.get("_kernel", "")
    except Exception:
        return None
    if t and k == kernel_name:
        return (t["total"], os.path.basename(f))
    return None


def host_cores():
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, int(float(q) / float(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = max(1, min(n, q // p))
        except Exception:
            pass
    return n


def launch_own_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no torch.distributed environment: start the N ranks ourselves (one process per
    device under torch.distributed.run, rendezvous on 127.0.0.1) and exit with their status.  Fails loudly when fewer than N devices
    are visible — a 1-GPU number must never be printed under --gpus N.  Returns only when there is nothing to launch (N = 1, or the
    ranks were already started by a launcher: WORLD_SIZE is set)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.dry_run:
        import torch

        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but %d HIP device(s) visible" % (args.gpus, have))
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    raise SystemExit(subprocess.call(cmd, env=env))


def rank_devices(torch, dist, rank, world, local_rank):
    """What every rank really runs on (rank 0 prints it): the world size the process group saw, the device of each rank."""
    if torch.cuda.is_available() and torch.cuda.device_count() > local_rank:
        p = torch.cuda.get_device_properties(local_rank)
        mine = {"rank": rank, "device": local_rank, "name": p.name, "arch": getattr(p, "gcnArchName", ""), "pid": os.getpid()}
    else:
        mine = {"rank": rank, "device": None, "name": "cpu (dry run)", "pid": os.getpid()}
    if world == 1:
        return [mine]
    out = [None] * world
    dist.all_gather_object(out, mine)
    return out


def dry_run(args, rank, world, local_rank):
    """--dry-run: everything of the N>1 run except the solves — process group (gloo or nccl), contiguous sharding (faster_amd/shard.py),
    the per-step gather of packed result records (zeroed, CPU tensors), barrier + max-over-ranks timing — and one JSON line without
    a `value`.  Touches neither the HIP library nor the oracle."""
    import torch
    import torch.distributed as dist

    from faster_amd import corridor, shard

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
    devices = rank_devices(torch, dist, rank, world, local_rank)
    strong = args.scaling == "strong" and world > 1
    pairs = min(args.pairs, 4096)
    whole, faces, _ = corridor.whole_batch(pairs, seed=3 if strong else 3 + 1000 * rank, n_seg=args.n_seg,
                                           p_choices=tuple(range(args.min_poly, args.max_poly + 1)))
    if strong:
        whole, faces = shard.shard_batch(whole, faces, rank, world)
    B = len(whole)
    per_rank = -(-pairs // world) if strong else B
    REC = 64 + 96 * args.n_seg  # fh_pack_results record
    PK = shard.gather_bytes_per_problem(args.gather, args.n_seg)  # what the gather moves per problem: packed record / 48-byte head / nothing
    wrec, srec = torch.zeros(per_rank * REC, dtype=torch.uint8), torch.zeros(per_rank * REC, dtype=torch.uint8)
    wrec[: B * REC] = rank + 1
    gather = None
    if world > 1 and PK:
        gather = [torch.zeros(world * per_rank * PK, dtype=torch.uint8) for _ in range(2)] if strong else (
            [[torch.zeros(per_rank * PK, dtype=torch.uint8) for _ in range(world)] for _ in range(2)] if rank == 0 else None)
    traffic = shard.gather_traffic(args.gather, args.n_seg, args.assume_pairs_per_s, world, strong)
    if world > 1 and not traffic["within_budget"] and not args.allow_over_budget:
        raise SystemExit("bench.py --dry-run: --gather %s at %.1f M pairs/s per GPU puts %.1f GB/s on every xGMI link (%.1f GB/s into one GPU); the stated "
                         "budget is %.1f GB/s per link (%.0f %% of %.1f GB/s one way).  Use --gather summaries / none, or --allow-over-budget"
                         % (args.gather, args.assume_pairs_per_s / 1e6, traffic["gather_GBps_per_link"], traffic["gather_GBps_into_busiest_gpu"],
                            traffic["budget_GBps_per_link"], 100 * shard.XGMI_GATHER_BUDGET_FRACTION, shard.XGMI_LINK_ONE_WAY_GBPS))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1 and PK:
            wp, sp = (wrec, srec) if args.gather == "records" else (shard.result_heads(wrec, per_rank, REC), shard.result_heads(srec, per_rank, REC))
            shard.gather_result_blocks(dist, wp, sp, gather, strong, rank)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ok = True
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        if not PK:
            ok = True
        elif strong:
            for r in range(world):
                lo, hi = shard.shard_range(pairs, r, world)
                ok &= bool((gather[0][r * per_rank * PK: (r * per_rank + hi - lo) * PK] == r + 1).all())
        elif rank == 0:
            ok = all(bool((gather[0][r] == r + 1).all()) for r in range(world))
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "trajectory solves/sec (whole+safe pairs) at N=%d, deg=3" % args.n_seg, "value": None,
                          "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "scaling": "strong" if strong else "weak",
                          "backend": args.backend if world > 1 else None, "devices": devices, "pairs_per_rank": B, "gather_ok": bool(ok), "gather": args.gather,
                          "gather_bytes_per_problem": PK, **{k: traffic[k] for k in ("gather_GBps_per_link", "gather_GBps_into_busiest_gpu",
                                                                                     "budget_GBps_per_link", "within_budget", "pairs_per_s_per_rank")},
                          "ms_per_step": 1e3 * elapsed / max(args.steps, 1)}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96, help="timed steps (the end of the timed region drains the pipelines: a longer region is closer to the steady state)")
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--pairs", type=int, default=32768, help="whole+safe pairs per step: per rank (weak scaling) or in total (strong); C4: 32768")
    ap.add_argument("--n-seg", type=int, default=10)
    ap.add_argument("--max-poly", type=int, default=6)
    ap.add_argument("--min-poly", type=int, default=2)
    ap.add_argument("--workload", choices=["c4", "c5"], default="c4",
                    help="c4 (default, the metric's configuration): synthetic corridors; c5: Monte-Carlo forest, corridors from the "
                         "voxel path search + ellipsoid decomposition front-end, N=15, <=8 polytopes (BASELINE config 5)")
    ap.add_argument("--front", choices=["device", "host"], default="device",
                    help="c5 only: where the corridors come from — the device front-end (fh_map_* path search + fh_corridor_batch_device, "
                         "default) or the CPU front-end (same results: tests/test_gpu_round2.py)")
    ap.add_argument("--c5-rule", choices=["reference", "c4"], default="reference",
                    help="c5 only: how R is chosen and what the safe corridor is — reference (default): FASTER's findIndexH / findIndexR on the "
                         "device (fh_set_pair_rule mode 1, Ra 4 m), up to 5 polytopes from the one that holds R, not pulled in: the `c5` record "
                         "of the default run; c4: SURVEY.md 8(d)'s synthetic pairing (R at half of the trajectory, 3 polytopes pulled in by 0.2 m)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--r-margin", type=float, default=0.05,
                    help="hand-off keeps R at least this far inside its safe corridor (FASTER decomposes the safe corridor around R); "
                         "negative: SURVEY.md 8(d) to the letter (R may fall outside the shrunk corridor)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs outside the timed region (solo launch, copies, flop rate)")
    ap.add_argument("--solo-only", action="store_true", help="of the legs outside the timed region run the solo launch only (A/B runs)")
    ap.add_argument("--pipeline", choices=["fused", "split"], default="fused",
                    help="fused: one launch per step, each unit of work is a pair taken through whole solve, hand-off and safe solve; "
                         "split: whole launch -> hand-off launch -> safe launch (same results)")
    ap.add_argument("--inflight", type=int, default=14,
                    help="independent pipelines (context + HIP stream + output buffers); step i runs on pipeline i %% inflight")
    ap.add_argument("--no-share", action="store_true", help="one wavefront per problem (fh_params.share = 0)")
    ap.add_argument("--wg-per-cu", type=int, default=0, help="resident solves per CU (fh_sched.workgroups_per_cu; 0: the library's default)")
    ap.add_argument("--e2e-child", action="store_true",
                    help="internal: only the PCIe-inclusive leg, in a process of its own with the HIP runtime's default number of hardware queues "
                         "(the parent's timed region wants 16 of them, this leg 4: DESIGN.md 6); prints that leg's record")
    ap.add_argument("--sched", default="", help="fh_sched fields of the timed pipelines, e.g. look_every=32,min_nodes=8 (experiments; no result depends on them)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend of the N>1 run: nccl (= RCCL over xGMI, default); gloo only with --dry-run")
    ap.add_argument("--waiting-workgroups", type=int, default=0,
                    help="fh_sched.waiting_workgroups of the timed pipelines: workgroups that keep waiting for frames of hard trees when the fresh "
                         "problems of their launch run out (0: the library's default, CUs / 64 for big batches)")
    ap.add_argument("--pair-outputs", action="store_true",
                    help="fh_sched.pair_outputs = 1 for the timed pipelines: the fused pair launch writes every safe problem (record + rows) to memory, as the "
                         "staged hand-off does; default off = the library's default: a safe problem is built in LDS and written only if it is shared")
    ap.add_argument("--full-results", action="store_true",
                    help="fh_sched.compact_results = 0 for the timed pipelines: every word of every 1600-byte fh_result is written; default: only the "
                         "coefficient rows the kernel is built for (1024 bytes at N = 10) — what a caller that streams batches sets")
    ap.add_argument("--gather", choices=["records", "summaries", "none"], default="summaries",
                    help="what the per-step batch gather of an N>1 run moves over RCCL/xGMI: summaries (default) — the 48-byte head of every "
                         "fh_result (solved, trials, status, counters, factor, dt, cost): what a planner that keeps its trajectories where they "
                         "were solved needs; records — every packed fh_result (64 + 96 N bytes): at 20 M pairs/s that is 42 GB/s per rank, the "
                         "same order as an xGMI link, so the N>1 number would measure the gather; none — results are consumed on their GPU")
    ap.add_argument("--assume-pairs-per-s", type=float, default=21.4e6,
                    help="--dry-run only: the per-GPU solve rate the gather's xGMI traffic is priced at (default: the measured 1-GPU C4 rate)")
    ap.add_argument("--allow-over-budget", action="store_true", help="--dry-run only: report a gather above the stated xGMI budget instead of failing")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, shard the batch and run the per-step gather of (zeroed) packed result records, but solve nothing: "
                         "checks the launcher and the N>1 plumbing where there is no GPU (tests/test_distributed_gloo.py); prints no `value`")
    args = ap.parse_args()
    if args.backend == "gloo" and not args.dry_run:
        raise SystemExit("bench.py: --backend gloo is for --dry-run only (the hot path has no CPU fallback)")

    # more hardware queues than the HIP default (4) so that the in-flight pipelines really run concurrently
    if not args.e2e_child:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    launch_own_ranks(args)  # (--gpus N > 1 outside torch.distributed.run: does not return)
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: refusing to report a %d-GPU number under "
                         "--gpus %d" % (args.gpus, world, world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world, local_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the hot path has no CPU fallback)")
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d wants device %d but only %d are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)
    devices = rank_devices(torch, dist, rank, world, local_rank)

    from faster_amd import abi, capi, corridor, shard

    N = args.n_seg
    strong = args.scaling == "strong" and world > 1
    seed = 3 if strong else 3 + 1000 * rank
    if args.workload == "c5":
        from faster_amd import build as fb, frontend

        fb.build_frontend()
        N = args.n_seg = 15
        args.max_poly = 8
        if args.front == "device":
            fctx, fmap = capi.Context(local_rank), capi.Map(local_rank)
            frontend.forest_batch(256, seed=seed + 2, n_seg=N, max_poly=8, front="device", ctx=fctx, vmap=fmap, device=local_rank, search="jps")  # allocations
            whole, faces, finfo = frontend.forest_batch(args.pairs, seed=seed + 2, n_seg=N, max_poly=8, front="device", ctx=fctx, vmap=fmap,
                                                        device=local_rank, search="jps")
            fctx.close()
            fmap.close()
        else:
            whole, faces, finfo = frontend.forest_batch(args.pairs, seed=seed + 2, n_seg=N, max_poly=8, search="jps")
    else:
        whole, faces, _ = corridor.whole_batch(args.pairs, seed=seed, n_seg=N, p_choices=tuple(range(args.min_poly, args.max_poly + 1)))
    total_pairs = len(whole)  # (c5: pairs without a path are dropped)
    if strong:  # one batch for the whole job: this rank's contiguous block of it
        whole, faces = shard.shard_batch(whole, faces, rank, world)
    B = len(whole)
    per_rank = -(-total_pairs // world) if strong else B
    safe_t = corridor.safe_templates(whole)
    max_faces = int(whole["face_off"][np.arange(B), whole["n_poly"]].max()) if B else 8
    if world > 1:  # one kernel instantiation / LDS carve on every rank
        mf = torch.tensor([max_faces], device=dev)
        dist.all_reduce(mf, op=dist.ReduceOp.MAX)
        max_faces = int(mf.item())

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    d_whole, d_faces = to_dev(whole), to_dev(faces)
    RES = abi.result_dtype.itemsize
    PACKED = capi.packed_result_size(N)
    GB = shard.gather_bytes_per_problem(args.gather, N)   # bytes per problem that the per-step gather moves (0: none)
    par = abi.default_params()
    if args.no_share:
        par["share"] = 0

    # `--inflight` independent pipelines, each with its own solver context, HIP stream and output buffers: step i runs on
    # pipeline i % inflight, so the straggler problems of one step overlap the bulk of the next step.  Every step does the
    # complete work on the complete batch.
    class Pipe:
        pass

    sched_kw = {k: v for k, v in (("workgroups_per_cu", args.wg_per_cu), ("waiting_workgroups", args.waiting_workgroups)) if v}
    sched_kw.update({kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.sched.split(",") if "=" in kv})

    def make_pipe(r_margin):
        pp = Pipe()
        pp.stream = torch.cuda.Stream(device=dev)
        pp.ctx = capi.Context(local_rank, pair_outputs=args.pair_outputs, compact_results=not args.full_results)  # (the library's defaults are lazy pair outputs, full records)
        pp.ctx.set_stream(pp.stream.cuda_stream)
        pp.ctx.set_params(par)
        pp.ctx.set_pair_margin(r_margin)
        if args.workload == "c5" and args.c5_rule == "reference":
            pp.ctx.set_pair_rule(mode=1, r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5)   # Ra, delta_H, delta_a: faster.yaml
        if sched_kw:
            pp.ctx.set_sched(**sched_kw)
        pp.d_safe = to_dev(safe_t)
        pp.d_sfaces = torch.zeros_like(d_faces)
        pp.d_wres = torch.zeros(per_rank * RES, dtype=torch.uint8, device=dev)  # (strong: padded to the largest shard)
        pp.d_sres = torch.zeros_like(pp.d_wres)
        pp.gather = None
        if world > 1 and GB:  # what crosses xGMI: PACKED records (fh_pack_results_device: no dead coefficient rows, 64 + 96 N bytes each) or their 48-byte heads
            if args.gather == "records":
                pp.d_wpack = torch.zeros(per_rank * PACKED, dtype=torch.uint8, device=dev)
                pp.d_spack = torch.zeros_like(pp.d_wpack)
            if strong:
                pp.gather = [torch.zeros(world * per_rank * GB, dtype=torch.uint8, device=dev) for _ in range(2)]
            elif rank == 0:
                pp.gather = [[torch.zeros(per_rank * GB, dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(2)]
        return pp

    if args.e2e_child:
        pp = make_pipe(args.r_margin)
        rec = e2e_leg(torch, dev, pp, whole, faces, safe_t, B, N, max_faces)
        rec["hardware_queues"] = os.environ.get("GPU_MAX_HW_QUEUES", "HIP default (4)")
        print(json.dumps(rec))
        pp.ctx.close()
        return
    pipes = [make_pipe(args.r_margin) for _ in range(max(1, args.inflight))]
    torch.cuda.synchronize()
    step_no = [0]

    # the hand-off of a pair: C4 (and --c5-rule c4): R at half of the trajectory, 3 polytopes pulled in by 0.2 m;
    # C5 by default: the reference's rule for R (set on the contexts above), up to 5 polytopes, not pulled in
    shrink, max_safe_poly = (0.0, 5) if (args.workload == "c5" and args.c5_rule == "reference") else (0.2, 3)

    def run_step(pp, fused):
        c = pp.ctx
        if fused:
            c.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, max_faces, 0.5, shrink, max_safe_poly, pp.d_wres.data_ptr(),
                                 pp.d_safe.data_ptr(), pp.d_sfaces.data_ptr(), pp.d_sres.data_ptr())
        else:
            c.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, max_faces, pp.d_wres.data_ptr())
            c.pair_glue_device(d_whole.data_ptr(), pp.d_wres.data_ptr(), d_faces.data_ptr(), B, 0.5, shrink, max_safe_poly, pp.d_safe.data_ptr(),
                               pp.d_sfaces.data_ptr())
            c.solve_batch_device(pp.d_safe.data_ptr(), pp.d_sfaces.data_ptr(), B, N, max_faces, pp.d_sres.data_ptr())

    def step():
        pp = pipes[step_no[0] % len(pipes)]
        step_no[0] += 1
        run_step(pp, args.pipeline == "fused")
        if world > 1 and GB:  # the batch gather over RCCL/xGMI: the packed record of every result of the step, or its 48-byte head
            if args.gather == "records":
                pp.ctx.pack_results_device(pp.d_wres.data_ptr(), per_rank, N, pp.d_wpack.data_ptr())
                pp.ctx.pack_results_device(pp.d_sres.data_ptr(), per_rank, N, pp.d_spack.data_ptr())
                with torch.cuda.stream(pp.stream):
                    shard.gather_result_blocks(dist, pp.d_wpack, pp.d_spack, pp.gather, strong, rank)
            else:
                with torch.cuda.stream(pp.stream):  # (the context launches on pp.stream: the strided copy is ordered behind the solve)
                    shard.gather_result_blocks(dist, shard.result_heads(pp.d_wres, per_rank, RES), shard.result_heads(pp.d_sres, per_rank, RES),
                                               pp.gather, strong, rank)

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
    share_stats = last.ctx.share_stats()
    timed_launch_info, timed_kname = last.ctx.last_launch()   # (of a launch of the timed region: the untimed launches below run alone)
    wres = last.d_wres.cpu().numpy().view(abi.result_dtype)[:B]
    sres = last.d_sres.cpu().numpy().view(abi.result_dtype)[:B]
    if args.pipeline == "fused" and not args.pair_outputs:
        # the timed launches built the safe problems on chip; their records and rows (face counts for the algorithmic bytes, the pinned
        # re-solve of the compute leg) come from ONE more launch, untimed, with complete pair outputs — the same results bit for bit
        for pp in {id(last): last, id(pipes[0]): pipes[0]}.values():
            pp.ctx.set_sched(pair_outputs=1, **sched_kw)
            run_step(pp, True)
            pp.ctx.sync()
            pp.ctx.set_sched(pair_outputs=0, **sched_kw)
        again = last.d_sres.cpu().numpy().view(abi.result_dtype)[:B]
        assert all(np.array_equal(again[f], sres[f]) for f in ("solved", "trials", "status", "factor", "dt", "cost")), "pair_outputs changed a result"
    safe_h = last.d_safe.cpu().numpy().view(abi.problem_dtype)
    sfaces_h = last.d_sfaces.cpu().numpy().view(abi.face_dtype)

    if rank == 0:
        fused = args.pipeline == "fused"
        launch_info, kname = timed_launch_info, timed_kname   # the instantiation and build that really ran, as rocprofv3 names it
        pairs_total = (total_pairs if strong else world * B) * args.steps
        value = pairs_total / elapsed
        bytes_whole = algorithmic_bytes(whole, N)
        active = safe_h["n_seg"] > 0
        bytes_safe = algorithmic_bytes(safe_h[active], N) + 24 * int((~active).sum())
        launches_per_step = 1 if fused else 2
        bytes_per_launch = (bytes_whole + bytes_safe) / launches_per_step
        avg_ms = float(np.mean(kernel_ms)) if len(kernel_ms) else float("nan")
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9  # GB/s
        traffic = measured_traffic(kname) or (None, None)
        out = {
            "metric": "trajectory solves/sec (whole+safe pairs) at N=%d, deg=3" % N,
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "devices": devices,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": ("C4: %d whole+safe paired solves per %s per step (N=%d segments, deg=3, <=%d polytopes whole / <=3 safe), "
                             "synthetic corridors (faster_amd/corridor.py seed 3), hand-off keeps R >= %.2f m inside its safe corridor"
                             % (args.pairs, "job" if strong else "GPU", N, args.max_poly, args.r_margin)) if args.workload == "c4" else
                            ("C5: %d whole+safe paired solves per %s per step in a random forest (20x20x3 m, 0.1 trees/m^2), corridors from "
                             "the voxel path search + ellipsoid decomposition front-end, N=15, <=8 polytopes; hand-off: %s"
                             % (total_pairs, "job" if strong else "GPU", "FASTER's rule for R on the device (Ra 4 m), <=5 polytopes from the one that "
                                "holds R" if args.c5_rule == "reference" else "SURVEY 8(d) pairing (R at half, 3 polytopes pulled in 0.2 m)")),
                "pairs_per_gpu": B,
                "pipeline": args.pipeline,
                "pipelines_in_flight": len(pipes),
                "pair_outputs": "complete (every safe problem written to memory)" if args.pair_outputs else "lazy (a safe problem is built on chip; written only when shared between workgroups)",
                "result_rows": "all 16" if args.full_results else "compact (the %d rows of the kernel's segment bucket)" % (6 if N <= 6 else (10 if N <= 10 else (15 if N <= 15 else 16))),
                "work_sharing": bool(par["share"]),
                "parallelism": (("one batch sharded x%d (contiguous blocks), RCCL all_gather of " % world if strong else "batch per GPU x%d, RCCL gather on rank 0 of " % world)
                                + {"records": "the packed result records (%d B each)" % PACKED, "summaries": "the 48-byte result heads (flags, counters, factor, dt, cost)",
                                   "none": "nothing (results stay on their GPU)"}[args.gather]
                                + ": %.2f GB/s per xGMI link, %.2f GB/s into the busiest GPU at the measured rate (budget %.1f GB/s per link)"
                                % tuple(shard.gather_traffic(args.gather, N, value / world, world, strong)[k] for k in
                                        ("gather_GBps_per_link", "gather_GBps_into_busiest_gpu", "budget_GBps_per_link"))) if world > 1 else "single GPU",
                **({"gather": shard.gather_traffic(args.gather, N, value / world, world, strong)} if world > 1 else {}),
                "whole_solved_frac": float(wres["solved"].mean()),
                "safe_solved_frac": float(sres["solved"].mean()),
                "mean_bnb_nodes_whole": float(wres["nodes"].mean()),
                "mean_bnb_nodes_safe": float(sres["nodes"].mean()),
                "mean_qp_iters_per_pair": float(wres["qp_iters"].mean() + sres["qp_iters"].mean()),
                "mean_trials_per_pair": float(wres["trials"].mean() + sres["trials"].mean()),
                **({"front_end": {"where": finfo["front"], "timing": finfo["front_timing"], "pairs_without_path": finfo["no_path"],
                                  "note": "corridor generation is outside the timed region (the metric is the solver's)"}}
                   if args.workload == "c5" else {}),
                "share_stats_last_launch": share_stats,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": kname,
                "mode": "timed region, %d launches in flight" % len(pipes),
                "achieved": achieved,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / (HBM_PEAK / 1e9),
                "traffic": traffic[0],
                "traffic_source": traffic[1],
                "wasted_x": (traffic[0] / bytes_per_launch) if traffic[0] else None,
                "launch": launch_info,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "avg_launch_ms": avg_ms,
                "launches_timed": int(len(kernel_ms)),
                "pipelines_in_flight": len(pipes),
                "aggregate_achieved": (bytes_whole + bytes_safe) * args.steps / elapsed / 1e9,
                "note": "the path is FP64-ALU / latency bound by construction (SURVEY.md 8(d): ~4-7 KB compulsory HBM bytes per pair). "
                        "Headline achieved / frac / avg_launch_ms: ONE launch alone on the GPU (`mode`), HIP events on the launching "
                        "stream, what rocprofv3 --kernel-trace reports for the same launch; `overlapped`: the launches of the timed "
                        "region, whose durations include time shared with the other launches in flight (not per-kernel evidence); "
                        "`aggregate_achieved`: all algorithmic bytes of the timed region over its wall time; `compute`: the FP64 flop "
                        "rate against the measured FP64 FMA peak",
            },
        }
        if world == 1 and not args.no_extra:
            out["roofline"]["solo"], solo_res = solo_leg(torch, pipes[0], run_step, fused, B, bytes_per_launch, launches_per_step)
            # the headline figure is the launch ALONE (per-kernel evidence); the overlapped launches of the timed region are labelled
            rf, so = out["roofline"], out["roofline"]["solo"]
            rf["overlapped"] = {"achieved": rf["achieved"], "frac": rf["frac"], "avg_launch_ms": rf["avg_launch_ms"], "unit": "GB/s",
                                "launches_timed": rf["launches_timed"], "pipelines_in_flight": rf["pipelines_in_flight"]}
            mean_solo_launch = float(np.mean(so["launch_ms_median"]))
            rf["achieved"], rf["frac"], rf["avg_launch_ms"], rf["mode"] = so["achieved"], so["frac"], mean_solo_launch, "one launch alone (solo leg)"
            if not args.wg_per_cu:
                # the build of the same kernel for two wavefronts per SIMD (fh_sched.workgroups_per_cu <= 8: all registers, no scratch):
                # what a caller who waits for ONE batch gets.  Not the kernel of the timed region: reported beside it, never as `achieved`.
                pipes[0].ctx.set_sched(workgroups_per_cu=8)
                lat, lat_res = solo_leg(torch, pipes[0], run_step, fused, B, bytes_per_launch, launches_per_step)
                pipes[0].ctx.set_sched(workgroups_per_cu=0)
                lat["same_results_as_the_throughput_build"] = bool(all(
                    np.array_equal(a[f], b[f]) for a, b in zip(solo_res, lat_res) for f in ("solved", "trials", "factor", "dt", "cost", "coeff", "assign")))
                lat["note"] = "fh_sched.workgroups_per_cu = 8: solve_kernel<N, PAIRS, 2> (two wavefronts per SIMD, 8 resident solves per CU)"
                rf["solo_two_wavefronts_per_simd"] = lat
            if args.solo_only:
                print(json.dumps(out))
                for pp in pipes:
                    pp.ctx.close()
                return
            out["roofline"]["compute"] = compute_leg(torch, dev, pipes[0], whole, faces, solo_res, N, max_faces,
                                                     out["roofline"]["solo"]["step_ms_median"], elapsed / args.steps, to_dev)
            out["e2e_with_copies"] = e2e_leg(torch, dev, pipes[0], whole, faces, safe_t, B, N, max_faces)
            out["e2e_with_copies"]["hardware_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
            if args.workload == "c4":
                # [r6] the same leg in a process of its own with the runtime's default number of hardware queues: what a caller that streams host
                # buffers (and does not keep fourteen launches in flight) gets — the queue count is read once per process
                try:
                    import subprocess

                    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
                    cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--e2e-child", "--pairs", str(args.pairs)], env=env,
                                        capture_output=True, text=True, timeout=240)
                    child = json.loads(cp.stdout.strip().splitlines()[-1])
                    out["e2e_with_copies"]["process_with_default_hardware_queues"] = {k: child[k] for k in (
                        "step_ms_median", "pairs_per_s", "pcie_GBps_both_ways", "lanes", "batches_streamed", "hardware_queues", "packed_equals_full_records")}
                except Exception as e:
                    out["e2e_with_copies"]["process_with_default_hardware_queues"] = {"error": repr(e)[:200]}
            if args.workload == "c4":
                lit_frac, lit_rate = literal_leg(torch, make_pipe, run_step, fused, abi, B, inflight=len(pipes))
                out["config"]["safe_solved_frac_literal_8d"] = lit_frac
                out["config"]["pairs_per_s_literal_8d"] = lit_rate
                out["single_replan_latency_ms"] = latency_leg(whole, faces)
                out["c5"] = c5_leg(torch, dev, local_rank, par, args.r_margin)
                out["replan_faithful"] = replan_leg(torch, dev, local_rank, par)
        if not args.no_cpu and world == 1:  # the CPU baseline is a property of the host: reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(whole, faces, safe_h, sfaces_h, args.cpu_seconds)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for pp in pipes:
        pp.ctx.close()


def solo_leg(torch, pp, run_step, fused, B, bytes_per_launch, launches_per_step, reps=7):
    """One batch alone on the GPU (no other launch in flight): median of `reps` fenced steps.  The per-launch HBM roofline
    fraction from these durations is not confounded by overlap."""
    import numpy as np

    from faster_amd import abi

    step_ms, launch_ms = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        pp.ctx.timing_reset()
        t = time.perf_counter()
        run_step(pp, fused)
        pp.ctx.sync()
        step_ms.append(1e3 * (time.perf_counter() - t))
        launch_ms.append(list(pp.ctx.timing_read()))
    launch = np.median(np.array(launch_ms), axis=0)
    med = float(np.median(step_ms))
    res = (pp.d_wres.cpu().numpy().view(abi.result_dtype)[:B].copy(), pp.d_sres.cpu().numpy().view(abi.result_dtype)[:B].copy())
    mean_launch = float(np.mean(launch))
    ach = bytes_per_launch / (mean_launch * 1e-3) / 1e9
    return {"step_ms_median": med, "launch_ms_median": [float(x) for x in launch], "repetitions": reps, "pairs_per_s": B / (med * 1e-3),
            "achieved": ach, "frac": ach / (HBM_PEAK / 1e9), "unit": "GB/s"}, res


def compute_leg(torch, dev, pp, whole, faces, solo_res, N, max_faces, solo_step_ms, steady_step_s, to_dev):
    """FP64 flop rate.  executed = the kernel's own count of the FP64 flops of every active-set iteration it ran
    (fh_result.kflops: useful lanes only, formula in fh_solve.hip.hpp qp_run); useful = the flops of ONE fixed-assignment QP at the
    winning factor with the winning assignment (the reference-equivalent minimal work of a trial), times the trials of the problem,
    measured by solving every solved problem again with its binaries pinned (fh_problem.pin) and the factor window shrunk to the
    winning factor.  peak = measured FP64 FMA rate of this device (fh_fp64_peak)."""
    import numpy as np

    from faster_amd import abi

    wres, sres = solo_res
    executed = 1e3 * (float(wres["kflops"].astype(np.float64).sum()) + float(sres["kflops"].astype(np.float64).sum()))
    # pinned re-solve of the whole problems (the safe problems live on the device: their records are read back)
    safe_h = pp.d_safe.cpu().numpy().view(abi.problem_dtype).copy()
    sfaces_h = pp.d_sfaces.cpu().numpy().view(abi.face_dtype).copy()
    useful = 0.0
    for probs, fcs, res in ((whole, faces, wres), (safe_h, sfaces_h, sres)):
        ok = (res["solved"] == 1) & (probs["n_seg"] > 0)
        if not ok.any():
            continue
        p = probs[ok].copy()
        r = res[ok]
        p["f_init"] = r["factor"]
        p["f_final"] = r["factor"]
        a = r["assign"].astype(np.int64)
        pins = np.zeros(len(p), dtype=np.uint64)
        for t in range(N):
            pins |= (np.where((a[:, t] >= 0) & (p["n_poly"] > 0), a[:, t] + 1, 0).astype(np.uint64) << np.uint64(4 * t))
        p["pin"][:, 0] = (pins & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        p["pin"][:, 1] = (pins >> np.uint64(32)).astype(np.uint32)
        d_p, d_f = to_dev(p), to_dev(fcs)
        d_r = torch.zeros(len(p) * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
        pp.ctx.solve_batch_device(d_p.data_ptr(), d_f.data_ptr(), len(p), N, max_faces, d_r.data_ptr())
        pp.ctx.sync()
        one = d_r.cpu().numpy().view(abi.result_dtype)
        useful += 1e3 * float((one["kflops"].astype(np.float64) * r["trials"]).sum())
    peak = pp.ctx.fp64_peak_tflops()
    ach_solo = executed / (solo_step_ms * 1e-3) / 1e12
    ach_steady = executed / steady_step_s / 1e12
    return {"executed_flops_per_step": executed, "useful_flops_per_step": useful, "useful_over_executed": useful / executed if executed else None,
            "achieved_tflops": ach_steady, "achieved_tflops_solo": ach_solo, "useful_tflops": useful / steady_step_s / 1e12,
            "measured_peak_tflops": peak, "spec_peak_tflops": FP64_VECTOR_SPEC_TFLOPS, "frac": ach_steady / peak if peak else None,
            "frac_solo": ach_solo / peak if peak else None, "unit": "TFLOP/s",
            "note": "executed: FP64 flops of all active-set iterations (useful lanes; a wave64 FP64 instruction occupies the SIMD for the full "
                    "wavefront, so lane utilisation — 30 of 64 lanes for N=10 — is not in this figure); useful: one fixed-assignment QP per trial — SURVEY 8(d)'s definition; since round 5 about two of five trials are "
                    "refuted at y = 0 without any QP, so executed can be BELOW this figure (useful_over_executed > 1)"}


def e2e_leg(torch, dev, pp, whole, faces, safe_t, B, N, max_faces, reps=5, batches=24, n_lanes=8):
    """PCIe-inclusive rate: problems and faces start in pinned host memory, both result arrays end there (never `value`).
    `serial`: one batch, one stream: H2D -> fused launch -> D2H of the full fh_result records (round 2's figure).
    headline: `batches` batches streamed on `n_lanes` lanes (context + stream), PACKED result records (fh_pack_results_device) copied back:
    the copies of one batch overlap the solves and copies of the others.  [r6] Eight lanes: with two (rounds 3-5) at most two launches were
    in flight and the leg measured the solve of a launch alone, not the link — in a process with the HIP default of 4 hardware queues 9.3 M
    pairs/s with two lanes against 13.6 M with eight (scripts/r6/e2e_lanes.py: 134.6 MB per batch both ways in 2.4 ms = 56 GB/s over PCIe).
    Inside THIS process GPU_MAX_HW_QUEUES is 16 (what the twelve pipelines of the timed region want: 23.3 M pairs/s against 19.3 M with 4), and
    with 16 queues the leg reads 9.3-9.7 M whatever its lane count."""
    import numpy as np

    from faster_amd import abi, capi

    RES = abi.result_dtype.itemsize
    PK = capi.packed_result_size(N)

    def pinned(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).pin_memory()

    # ---- serial, full records ----
    h_whole, h_faces, h_safe = pinned(whole), pinned(faces), pinned(safe_t)
    h_wres = torch.zeros(B * RES, dtype=torch.uint8).pin_memory()
    h_sres = torch.zeros(B * RES, dtype=torch.uint8).pin_memory()
    d_whole = torch.empty_like(h_whole, device=dev)
    d_faces = torch.empty_like(h_faces, device=dev)
    ms = []
    for _ in range(reps + 1):
        torch.cuda.synchronize()
        t = time.perf_counter()
        with torch.cuda.stream(pp.stream):
            d_whole.copy_(h_whole, non_blocking=True)
            d_faces.copy_(h_faces, non_blocking=True)
            pp.d_safe.copy_(h_safe, non_blocking=True)
            pp.ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, max_faces, 0.5, 0.2, 3, pp.d_wres.data_ptr(),
                                      pp.d_safe.data_ptr(), pp.d_sfaces.data_ptr(), pp.d_sres.data_ptr())
            h_wres.copy_(pp.d_wres[: B * RES], non_blocking=True)
            h_sres.copy_(pp.d_sres[: B * RES], non_blocking=True)
        pp.stream.synchronize()
        ms.append(1e3 * (time.perf_counter() - t))
    serial = float(np.median(ms[1:]))
    serial_bytes = h_whole.numel() + h_faces.numel() + h_safe.numel() + 2 * B * RES

    # ---- streamed: whole batches on n_lanes lanes, packed records back ----
    class Lane:
        pass

    lanes = []
    for k in range(n_lanes):
        ln = Lane()
        ln.stream = torch.cuda.Stream(device=dev)
        ln.ctx = capi.Context(dev.index or 0)
        ln.ctx.set_stream(ln.stream.cuda_stream)
        ln.ctx.set_pair_margin(0.05)
        ln.d_whole, ln.d_faces = torch.empty_like(h_whole, device=dev), torch.empty_like(h_faces, device=dev)
        ln.d_safe, ln.d_sfaces = torch.empty_like(h_safe, device=dev), torch.zeros(h_faces.numel(), dtype=torch.uint8, device=dev)
        ln.d_wres = torch.zeros(B * RES, dtype=torch.uint8, device=dev)
        ln.d_sres = torch.zeros_like(ln.d_wres)
        ln.d_wpack = torch.zeros(B * PK, dtype=torch.uint8, device=dev)
        ln.d_spack = torch.zeros_like(ln.d_wpack)
        ln.h_wpack = torch.zeros(B * PK, dtype=torch.uint8).pin_memory()
        ln.h_spack = torch.zeros(B * PK, dtype=torch.uint8).pin_memory()
        lanes.append(ln)

    def one_batch(ln):
        with torch.cuda.stream(ln.stream):
            ln.d_whole.copy_(h_whole, non_blocking=True)
            ln.d_faces.copy_(h_faces, non_blocking=True)
            ln.d_safe.copy_(h_safe, non_blocking=True)
            ln.ctx.solve_pairs_device(ln.d_whole.data_ptr(), ln.d_faces.data_ptr(), B, N, max_faces, 0.5, 0.2, 3, ln.d_wres.data_ptr(),
                                      ln.d_safe.data_ptr(), ln.d_sfaces.data_ptr(), ln.d_sres.data_ptr())
            ln.ctx.pack_results_device(ln.d_wres.data_ptr(), B, N, ln.d_wpack.data_ptr())
            ln.ctx.pack_results_device(ln.d_sres.data_ptr(), B, N, ln.d_spack.data_ptr())
            ln.h_wpack.copy_(ln.d_wpack, non_blocking=True)
            ln.h_spack.copy_(ln.d_spack, non_blocking=True)

    for ln in lanes:  # allocations
        one_batch(ln)
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(batches):
            one_batch(lanes[k % n_lanes])
        for ln in lanes:
            ln.stream.synchronize()
        ms.append(1e3 * (time.perf_counter() - t) / batches)
    med = float(np.median(ms))
    # the packed records are the full ones without the dead rows
    got = capi.unpack_results(lanes[0].h_wpack.numpy(), B, N)
    want = h_wres.numpy().view(abi.result_dtype)
    same = all(np.array_equal(got[f], want[f]) for f in ("solved", "trials", "factor", "dt", "cost", "coeff", "assign"))
    nbytes = h_whole.numel() + h_faces.numel() + h_safe.numel() + 2 * B * PK
    for ln in lanes:
        ln.ctx.close()
    return {"step_ms_median": med, "repetitions": reps, "pairs_per_s": B / (med * 1e-3), "bytes_over_pcie_per_step": int(nbytes),
            "batches_streamed": batches, "lanes": n_lanes, "pcie_GBps_both_ways": nbytes / (med * 1e-3) / 1e9, "packed_record_bytes": PK, "packed_equals_full_records": bool(same),
            "serial_full_records": {"step_ms_median": serial, "pairs_per_s": B / (serial * 1e-3), "bytes_over_pcie_per_step": int(serial_bytes)},
            "note": "pinned host memory; EVERY batch crosses PCIe both ways. headline: %d batches streamed on %d lanes (context + stream each): "
                    "H2D problems+faces+safe templates -> fused pair launch -> pack -> D2H of the packed result records (%d B instead of "
                    "%d), so that the copies of one batch overlap the solve of the next; serial_full_records: one batch at a time on one "
                    "stream with the full records (round 2's figure)" % (batches, n_lanes, PK, RES)}


def replan_leg(torch, dev, local_rank, par, pairs=65536, reps=3):
    """One Faster::replan per pair with FASTER's own parameters (faster.yaml: N_whole = N_safe = 6, max_poly_whole = max_poly_safe = 3,
    dist_max_vertexes 1.5 m, Ra 4 m, delta_H 1, delta_a 0.5) and its own steps, device-resident from the point cloud and the
    start/goal pairs to the committed plans, outside the C4 timed region: map, jump point search, JPS_in (the path inside the sphere Ra),
    createMoreVertexes / deleteVertexes, whole corridor, problem records (E = G or the last vertex), whole solve, the safe corridor
    decomposed around R against unknown + occupied space (unknown space modelled: farther than Ra from the start), safe solve,
    appendToPlan.  Median of `reps` fenced passes per stage."""
    import numpy as np

    from faster_amd import abi, capi, corridor, frontend

    N, max_poly, r_known, drone_r, decomp_r, fpp, max_states = 6, 3, 4.0, 0.3, 0.05, 96, 512
    res, infl, zmax = 0.2, 0.3, 3.0
    cloud, cells, center, starts, goals, rng = frontend.forest_queries(pairs, 7, return_rng=True)
    B = pairs
    whole = abi.make_problems(B)
    whole["n_seg"], whole["force_final_pos"], whole["dc"] = N, 1, 0.01
    whole["v_max"], whole["a_max"], whole["j_max"] = 5.0, 5.0, 8.0
    whole["f_init"], whole["f_final"], whole["f_inc"] = 1.0, 10.0, 1.0
    u = goals - starts
    u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-9)
    whole["x0"][:, 0:3] = starts
    whole["x0"][:, 3:6] = u * rng.uniform(0, 1.5, size=(B, 1))
    tmpl = corridor.safe_templates(whole)
    tmpl["n_seg"] = N

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    ctx, vmap = capi.Context(local_rank), capi.Map(local_rank)
    mp = 16  # vertices of JPS_in kept: the WHOLE path inside the sphere — the march towards unknown space runs along all of it
             # (faster.cpp:446-452); the whole corridor uses its first max_poly legs (deleteVertexes, :390-392: fh_corridor_batch_device)
    d_cloud, d_starts, d_goals = to_dev(cloud), to_dev(starts), to_dev(goals)
    d_whole_t, d_tmpl = to_dev(whole), to_dev(tmpl)
    d_whole, d_safe = torch.zeros_like(d_whole_t), torch.zeros_like(d_tmpl)
    f64, i32 = torch.float64, torch.int32
    d_paths, d_np, d_ex = torch.zeros((B, mp, 3), dtype=f64, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=torch.int64, device=dev)
    FB = abi.face_dtype.itemsize
    d_wf, d_sf = torch.zeros(B * fpp * FB, dtype=torch.uint8, device=dev), torch.zeros(B * fpp * FB, dtype=torch.uint8, device=dev)
    d_off, d_npoly, d_last = torch.zeros((B, 9), dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros((B, 3), dtype=f64, device=dev)
    RES = abi.result_dtype.itemsize
    d_wr, d_sr = torch.zeros(B * RES, dtype=torch.uint8, device=dev), torch.zeros(B * RES, dtype=torch.uint8, device=dev)
    d_plans = torch.zeros(B * max_states * abi.state_dtype.itemsize, dtype=torch.uint8, device=dev)
    d_counts, d_k = torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev)
    ctx.set_params(par)
    ctx.set_pair_rule(mode=1, r_known=r_known, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
    vmap.set_search("jps")
    vmap.set_sphere(r_known)
    stages = {"map": [], "path_search": [], "whole_corridor": [], "whole_solve": [], "safe_corridor": [], "safe_solve": [], "append_plans": []}

    def timed(name, fn, sync):
        torch.cuda.synchronize()
        t = time.perf_counter()
        fn()
        sync()
        stages[name].append(1e3 * (time.perf_counter() - t))

    dims = origin = None
    try:
        for _ in range(reps + 1):
            d_whole.copy_(d_whole_t)
            d_safe.copy_(d_tmpl)
            timed("map", lambda: vmap.read_device(d_cloud.data_ptr(), len(cloud), cells, res, center, 0.0, zmax, infl), vmap.sync)
            timed("path_search", lambda: vmap.plan_batch_device(d_starts.data_ptr(), d_goals.data_ptr(), B, mp, d_paths.data_ptr(), d_np.data_ptr(),
                                                                d_ex.data_ptr(), 1.5, 0), vmap.sync)
            if dims is None:
                dims, origin = vmap.dims()

            def corridor_stage():
                ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), d_paths.data_ptr(), d_np.data_ptr(), B, mp, max_poly, fpp, d_wf.data_ptr(),
                                          d_off.data_ptr(), d_npoly.data_ptr(), d_last.data_ptr(), decomp_r, 0.0)
                ctx.corridor_problems_device(d_np.data_ptr(), d_last.data_ptr(), d_goals.data_ptr(), d_wf.data_ptr(), d_off.data_ptr(), d_npoly.data_ptr(),
                                             B, fpp, N, d_whole.data_ptr())
            timed("whole_corridor", corridor_stage, ctx.sync)
            timed("whole_solve", lambda: ctx.solve_batch_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, d_wr.data_ptr()), ctx.sync)
            timed("safe_corridor", lambda: ctx.safe_corridor_batch_device(d_whole.data_ptr(), d_wr.data_ptr(), d_paths.data_ptr(), d_np.data_ptr(), mp,
                                                                          d_goals.data_ptr(), d_cloud.data_ptr(), len(cloud), origin, res, dims, B, 0.5,
                                                                          max_poly, (2.0, 2.0, 1.0), decomp_r, 0.0, fpp, N, d_safe.data_ptr(),
                                                                          d_sf.data_ptr()), ctx.sync)
            timed("safe_solve", lambda: ctx.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, fpp, d_sr.data_ptr()), ctx.sync)
            timed("append_plans", lambda: ctx.append_plans_device(d_whole.data_ptr(), d_wr.data_ptr(), d_safe.data_ptr(), d_sr.data_ptr(), B, 0.5,
                                                                  max_states, d_plans.data_ptr(), d_counts.data_ptr(), d_k.data_ptr()), ctx.sync)
        wres, sres = d_wr.cpu().numpy().view(abi.result_dtype).copy(), d_sr.cpu().numpy().view(abi.result_dtype).copy()
        wprob, safe = d_whole.cpu().numpy().view(abi.problem_dtype).copy(), d_safe.cpu().numpy().view(abi.problem_dtype).copy()
        counts = d_counts.cpu().numpy().copy()
        pops = int(d_ex.sum().item())
        # ---- [r6] the same chain with its stages OVERLAPPED: three batches in flight, each on a lane of its own (a map, a context and a
        # stream: every call of the chain is asynchronous on its lane's stream), so that the path search of one batch — issue-bound, 73 % of
        # a staged replan — runs beside the solves and decompositions of the others.  Every batch is the whole chain from the cloud to
        # the committed plans; the plans of a lane are compared with the staged run's.
        pipelined = None
        try:
            class RLane:
                pass

            rl = []
            for k in range(3):
                ln = RLane()
                ln.stream = torch.cuda.Stream(device=dev)
                ln.ctx, ln.vmap = capi.Context(local_rank), capi.Map(local_rank)
                ln.ctx.set_stream(ln.stream.cuda_stream); ln.vmap.set_stream(ln.stream.cuda_stream)
                ln.ctx.set_params(par)
                ln.ctx.set_pair_rule(mode=1, r_known=r_known, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
                ln.vmap.set_search("jps"); ln.vmap.set_sphere(r_known)
                for nm, t in (("whole", d_whole), ("safe", d_safe), ("paths", d_paths), ("np", d_np), ("ex", d_ex), ("wf", d_wf), ("sf", d_sf), ("off", d_off),
                              ("npoly", d_npoly), ("last", d_last), ("wr", d_wr), ("sr", d_sr), ("plans", d_plans), ("counts", d_counts), ("k", d_k)):
                    setattr(ln, nm, torch.zeros_like(t))
                rl.append(ln)

            def issue(ln):
                with torch.cuda.stream(ln.stream):
                    ln.whole.copy_(d_whole_t, non_blocking=True)
                    ln.safe.copy_(d_tmpl, non_blocking=True)
                ln.vmap.read_device(d_cloud.data_ptr(), len(cloud), cells, res, center, 0.0, zmax, infl)
                ln.vmap.plan_batch_device(d_starts.data_ptr(), d_goals.data_ptr(), B, mp, ln.paths.data_ptr(), ln.np.data_ptr(), ln.ex.data_ptr(), 1.5, 0)
                ln.ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), ln.paths.data_ptr(), ln.np.data_ptr(), B, mp, max_poly, fpp, ln.wf.data_ptr(),
                                             ln.off.data_ptr(), ln.npoly.data_ptr(), ln.last.data_ptr(), decomp_r, 0.0)
                ln.ctx.corridor_problems_device(ln.np.data_ptr(), ln.last.data_ptr(), d_goals.data_ptr(), ln.wf.data_ptr(), ln.off.data_ptr(), ln.npoly.data_ptr(),
                                                B, fpp, N, ln.whole.data_ptr())
                ln.ctx.solve_batch_device(ln.whole.data_ptr(), ln.wf.data_ptr(), B, N, fpp, ln.wr.data_ptr())
                ln.ctx.safe_corridor_batch_device(ln.whole.data_ptr(), ln.wr.data_ptr(), ln.paths.data_ptr(), ln.np.data_ptr(), mp, d_goals.data_ptr(),
                                                  d_cloud.data_ptr(), len(cloud), origin, res, dims, B, 0.5, max_poly, (2.0, 2.0, 1.0), decomp_r, 0.0, fpp, N,
                                                  ln.safe.data_ptr(), ln.sf.data_ptr())
                ln.ctx.solve_batch_device(ln.safe.data_ptr(), ln.sf.data_ptr(), B, N, fpp, ln.sr.data_ptr())
                ln.ctx.append_plans_device(ln.whole.data_ptr(), ln.wr.data_ptr(), ln.safe.data_ptr(), ln.sr.data_ptr(), B, 0.5, max_states, ln.plans.data_ptr(),
                                           ln.counts.data_ptr(), ln.k.data_ptr())

            for ln in rl:  # allocations, jump tables, first-launch set-up: untimed
                issue(ln)
            torch.cuda.synchronize()
            batches = 9
            t = time.perf_counter()
            for b in range(batches):
                issue(rl[b % len(rl)])
            torch.cuda.synchronize()
            el = time.perf_counter() - t
            same_counts = all(bool(torch.equal(ln.counts, d_counts)) for ln in rl)
            same_plans = all(bool(torch.equal(ln.plans, d_plans)) for ln in rl)
            pipelined = {"lanes": len(rl), "batches": batches, "ms_per_batch": 1e3 * el / batches, "replans_per_s": batches * B / el,
                         "plans_identical_to_the_staged_run": bool(same_counts and same_plans),
                         "note": "the whole chain of a batch (map, path search, corridors, both solves, appendToPlan) issued asynchronously on the stream of "
                                 "its lane, three lanes in flight: throughput is bound by what the stages share of the device, not by their sum"}
            for ln in rl:
                ln.vmap.close(); ln.ctx.close()
        except Exception as e:  # (the staged figures stand on their own)
            pipelined = {"error": repr(e)[:300]}
        # ---- the same pairs with unknown space as an INPUT (fh_pair_rule mode 2): the voxels of the map's lattice that a vehicle which has
        # explored a dozen spheres has not seen.  Map, paths and whole trajectories do not depend on it: the three stages that do, again.
        iz, iy, ix = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
        cen = np.stack([(ix + 0.5) * res + origin[0], (iy + 0.5) * res + origin[1], (iz + 0.5) * res + origin[2]], axis=-1)
        seen = np.zeros(iz.shape, dtype=bool)
        for c in rng.uniform([1, 1, 1.5], [19, 19, 1.5], size=(16, 3)):
            seen |= np.linalg.norm(cen - c, axis=-1) < rng.uniform(2.0, 3.5)
        d_flags = to_dev((~seen).astype(np.uint8))
        ctx.set_pair_rule(mode=2, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
        ctx.set_unknown_grid_device(d_flags.data_ptr(), origin, res, dims)
        fpp2 = 192  # (polytopes against real unknown voxels have more rows: 3 x 64)
        d_sf2 = torch.zeros(B * fpp2 * FB, dtype=torch.uint8, device=dev)
        stages2 = {"safe_corridor": [], "safe_solve": [], "append_plans": []}
        stages_sphere, stages = stages, stages2
        for _ in range(reps + 1):
            d_safe.copy_(d_tmpl)
            timed("safe_corridor", lambda: ctx.safe_corridor_batch_device(d_whole.data_ptr(), d_wr.data_ptr(), d_paths.data_ptr(), d_np.data_ptr(), mp,
                                                                          d_goals.data_ptr(), d_cloud.data_ptr(), len(cloud), origin, res, dims, B, 0.5,
                                                                          max_poly, (2.0, 2.0, 1.0), decomp_r, 0.0, fpp2, N, d_safe.data_ptr(),
                                                                          d_sf2.data_ptr()), ctx.sync)
            timed("safe_solve", lambda: ctx.solve_batch_device(d_safe.data_ptr(), d_sf2.data_ptr(), B, N, fpp2, d_sr.data_ptr()), ctx.sync)
            timed("append_plans", lambda: ctx.append_plans_device(d_whole.data_ptr(), d_wr.data_ptr(), d_safe.data_ptr(), d_sr.data_ptr(), B, 0.5,
                                                                  max_states, d_plans.data_ptr(), d_counts.data_ptr(), d_k.data_ptr()), ctx.sync)
        sres2, safe2, counts2 = d_sr.cpu().numpy().view(abi.result_dtype).copy(), d_safe.cpu().numpy().view(abi.problem_dtype).copy(), d_counts.cpu().numpy().copy()
        need2 = safe2["n_seg"] > 0
        # [r5] the same decisions inside ONE launch: the fused pair kernel with rule mode 2 (whole solve -> H, R and "is a safe trajectory
        # needed" against the unknown voxels -> safe solve in the polytopes of the whole corridor from the one that holds R; the corridor
        # decomposed around R stays the staged path above)
        stages2["fused_whole_handoff_safe"] = []
        d_wr3, d_sr3, d_safe3, d_sf3 = torch.zeros_like(d_wr), torch.zeros_like(d_sr), torch.zeros_like(d_tmpl), torch.zeros_like(d_wf)
        ctx.set_pair_margin(0.0)
        for _ in range(reps + 1):
            d_safe3.copy_(d_tmpl)
            timed("fused_whole_handoff_safe", lambda: ctx.solve_pairs_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, 0.5, 0.0, max_poly, d_wr3.data_ptr(),
                                                                             d_safe3.data_ptr(), d_sf3.data_ptr(), d_sr3.data_ptr()), ctx.sync)
        ctx.set_pair_margin(-1.0)
        fused_kernel = ctx.last_launch()[1]
        safe3, sres3 = d_safe3.cpu().numpy().view(abi.problem_dtype), d_sr3.cpu().numpy().view(abi.result_dtype)
        need3 = safe3["n_seg"] > 0
        both = need2 & need3
        stages = stages_sphere
        med2 = {k: float(np.median(v[1:])) for k, v in stages2.items()}
        unknown_input = {"unknown_voxel_frac": float((~seen).mean()), "stages_ms": med2, "pairs_needing_a_safe_trajectory": int(need2.sum()),
                         "safe_solved_frac": float(sres2["solved"][need2].mean()) if need2.any() else None,
                         "plans_committed_frac": float((counts2 > 0).mean()),
                         "fused": {"kernel": fused_kernel, "ms": med2["fused_whole_handoff_safe"], "pairs_needing_a_safe_trajectory": int(need3.sum()),
                                   "same_r_as_the_staged_chain": float(np.all(np.abs(safe3["x0"][both] - safe2["x0"][both]) <= 1e-12, axis=1).mean()) if both.any() else None,
                                   "safe_solved_frac": float(sres3["solved"][need3].mean()) if need3.any() else None,
                                   "label": "OCCUPIED-SPACE corridor, NOT FASTER's safe corridor (its polytopes are the whole corridor's: decomposed against "
                                            "occupied points only, not against unknown + occupied space as faster.cpp:493-499) - not an alternative to the "
                                            "staged chain's safe_corridor + safe_solve, a different and easier problem",
                                   "note": "fh_solve_pairs_device with rule mode 2: whole solve + hand-off + safe solve of all pairs in one launch "
                                           "(stages_ms.fused_whole_handoff_safe) — against whole_solve above + safe_corridor + safe_solve of the staged "
                                           "chain; its safe corridor is the run of polytopes of the whole corridor from the one that holds R"},
                         "note": "fh_set_unknown_grid_device + fh_pair_rule mode 2: findIndexH and the march of getFirstCollisionJPS ask for the "
                                 "nearest unknown voxel of the grid (exact, as the reference's kd-tree), the safe corridor is decomposed against "
                                 "[unknown voxels | occupied points]; map, paths and whole solves as above"}
    finally:
        vmap.close()
        ctx.close()
    med = {k: float(np.median(v[1:])) for k, v in stages.items()}
    total_ms = sum(med.values())
    have = wprob["n_seg"] > 0
    need = safe["n_seg"] > 0
    return {"workload": "one Faster::replan per pair, FASTER's own parameters (N 6, max_poly 3, Ra 4 m): %d start/goal pairs in a random forest "
                        "(20x20x3 m, 0.1 trees/m^2), %d with a path; device-resident from the cloud and the queries to the committed plans, one "
                        "stage at a time" % (pairs, int(have.sum())),
            "stages_ms": med, "replans_per_s": B / (total_ms * 1e-3), "total_ms": total_ms, "heap_pops_of_the_path_search": pops,
            "whole_solved_frac": float(wres["solved"][have].mean()), "pairs_needing_a_safe_trajectory": int(need.sum()),
            "safe_solved_frac": float(sres["solved"][need].mean()) if need.any() else None,
            "plans_committed_frac": float((counts > 0).mean()), "mean_plan_states": float(counts[counts > 0].mean()) if (counts > 0).any() else 0.0,
            "unknown_space_as_an_input": unknown_input,
            "pipelined": pipelined,
            "note": "unknown space is MODELLED (a batch has no mapper): everything farther than Ra from the start — distance queries use "
                    "Ra - |p - A|, the decomposition sees the voxels of the map's grid out there; path_search includes building the jump tables "
                    "of the map; whole_corridor includes fh_corridor_problems_device (E = G or the last vertex)"}


def c5_leg(torch, dev, local_rank, par, r_margin, pairs=65536, reps=3):
    """BASELINE config C5 as a record of the N=1 line (outside the C4 timed region): 65536 start/goal pairs in one random forest,
    corridors from the DEVICE front-end (voxel map + path search + ellipsoid decomposition), then one fused whole+safe launch over
    all of them with `solve_kernel<15, true, 2>` (LDS admits 5 solves per CU: the build for two wavefronts per SIMD); everything stays in HBM between the two.  Median of `reps` fenced steps."""
    import numpy as np

    from faster_amd import abi, capi, corridor, frontend

    N = 15
    fctx, fmap = capi.Context(local_rank), capi.Map(local_rank)
    try:
        frontend.forest_batch(256, seed=5, n_seg=N, max_poly=8, front="device", ctx=fctx, vmap=fmap, device=local_rank, search="jps")  # allocations
        whole, faces, finfo = frontend.forest_batch(pairs, seed=5, n_seg=N, max_poly=8, front="device", ctx=fctx, vmap=fmap, device=local_rank, search="jps")
    finally:
        fmap.close()
    B = len(whole)
    tmpl = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())

    def to_dev(a):
        return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)

    d_whole, d_faces, d_safe = to_dev(whole), to_dev(faces), to_dev(tmpl)
    d_sf = torch.zeros_like(d_faces)
    d_wr = torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device=dev)
    d_sr = torch.zeros_like(d_wr)
    fctx.set_params(par)
    fctx.set_pair_margin(r_margin)

    def run(mode, shrink, max_safe_poly):
        fctx.set_pair_rule(mode=mode, r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5)   # Ra, delta_H, delta_a: faster.yaml
        d_safe.copy_(to_dev(tmpl))
        ms = []
        for k in range(reps + 1):
            torch.cuda.synchronize()
            t = time.perf_counter()
            fctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, shrink, max_safe_poly, d_wr.data_ptr(), d_safe.data_ptr(),
                                    d_sf.data_ptr(), d_sr.data_ptr())
            fctx.sync()
            ms.append(1e3 * (time.perf_counter() - t))
        w, s_ = d_wr.cpu().numpy().view(abi.result_dtype).copy(), d_sr.cpu().numpy().view(abi.result_dtype).copy()
        sp = d_safe.cpu().numpy().view(abi.problem_dtype)
        return float(np.median(ms[1:])), w, s_, int((sp["n_seg"] > 0).sum())

    med_half, wres_half, sres_half, live_half = run(0, 0.2, 3)   # C4's synthetic pairing applied to the forest corridors
    med, wres, sres, live = run(1, 0.0, 5)                       # FASTER's own rule for R; safe corridor long enough to reach from R to H
    fctx.close()
    # [r6] ... and the same launch with several batches IN FLIGHT (what the C4 headline measures for its configuration): a lane = a context,
    # a stream, its own safe problems and results; every batch is the complete launch over all B pairs; results compared with the launch alone
    inflight = None
    try:
        class CLane:
            pass

        cl = []
        for k in range(int(os.environ.get("FH_BENCH_C5_LANES", "4"))):
            ln = CLane()
            ln.stream = torch.cuda.Stream(device=dev)
            ln.ctx = capi.Context(local_rank)
            ln.ctx.set_stream(ln.stream.cuda_stream)
            ln.ctx.set_params(par)
            ln.ctx.set_pair_margin(r_margin)
            ln.ctx.set_pair_rule(mode=1, r_known=4.0, drone_radius=0.3, delta_h=1.0, delta_a=0.5)
            ln.d_safe, ln.d_sf, ln.d_wr, ln.d_sr = to_dev(tmpl), torch.zeros_like(d_faces), torch.zeros_like(d_wr), torch.zeros_like(d_wr)
            cl.append(ln)

        def c5_issue(ln):
            ln.ctx.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.0, 5, ln.d_wr.data_ptr(), ln.d_safe.data_ptr(), ln.d_sf.data_ptr(),
                                      ln.d_sr.data_ptr())

        for ln in cl:
            c5_issue(ln)
        torch.cuda.synchronize()
        nb = 3 * len(cl)
        t = time.perf_counter()
        for b in range(nb):
            c5_issue(cl[b % len(cl)])
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        w4 = cl[0].d_wr.cpu().numpy().view(abi.result_dtype)
        s4 = cl[0].d_sr.cpu().numpy().view(abi.result_dtype)
        same = all(np.array_equal(a[f], b[f]) for a, b in ((w4, wres), (s4, sres)) for f in ("solved", "trials", "factor", "dt", "cost", "coeff", "assign"))
        inflight = {"lanes": len(cl), "batches": nb, "ms_per_batch": 1e3 * el / nb, "pairs_per_s": nb * B / el, "same_results_as_the_launch_alone": bool(same)}
        for ln in cl:
            ln.ctx.close()
    except Exception as e:  # (the launch alone stands on its own)
        inflight = {"error": repr(e)[:300]}
    ft = finfo["front_timing"]
    front_s = ft["map_s"] + ft["path_search_s"] + ft["decomposition_s"]
    return {"workload": "C5: %d whole+safe pairs (of %d queries with a path) in a random forest (20x20x3 m, 0.1 trees/m^2), N=15, <=8 polytopes, "
                        "corridors from the device front-end; one fused launch alone on the GPU" % (B, pairs),
            "kernel": "fh::solve_kernel<15, true, 2>", "pairs": B, "step_ms_median": med, "pairs_per_s": B / (med * 1e-3), "repetitions": reps,
            "batches_in_flight": inflight,
            "front_end": {"map_s": ft["map_s"], "path_search_s": ft["path_search_s"], "decomposition_s": ft["decomposition_s"],
                          "corridors_per_s": pairs / front_s, "expansions": ft["expansions"],
                          "path_search": "jump point search in jps3d's own order (fh_map_set_search 1): FASTER's exact vertex lists; the jump "
                                         "tables of the map are built inside path_search_s"},
            "front_end_plus_solver_pairs_per_s": B / (front_s + med * 1e-3),
            "whole_solved_frac": float(wres["solved"].mean()),
            "safe_problems": live, "safe_solved_frac": float(sres["solved"].sum() / max(live, 1)),
            "mean_qp_iters_per_pair": float(wres["qp_iters"].mean() + sres["qp_iters"].mean()),
            "max_faces": mf,
            "hand_off": "R by FASTER's findIndexH / findIndexR on the device (fh_set_pair_rule mode 1: unknown space = farther than Ra = 4 m from "
                        "the start, drone_radius 0.3 m, delta_H 1.0, delta_a 0.5; pairs whose whole trajectory stays in known space need no safe "
                        "trajectory, faster.cpp:462-466); safe corridor = the polytopes of the whole corridor from the one that holds R, up to 5 "
                        "of them, not pulled in (they are <= 1.5 m long each, dist_max_vertexes: five reach from R to the unknown boundary, "
                        "which the reference covers with <= 3 longer polytopes decomposed anew around R..M, faster.cpp:475-499)",
            "r_at_half_of_the_trajectory": {"step_ms_median": med_half, "pairs_per_s": B / (med_half * 1e-3), "safe_problems": live_half,
                                            "safe_solved_frac": float(sres_half["solved"].sum() / max(live_half, 1)),
                                            "mean_qp_iters_per_pair": float(wres_half["qp_iters"].mean() + sres_half["qp_iters"].mean()),
                                            "note": "C4's synthetic pairing (SURVEY.md 8(d): R = the sample at half of the whole trajectory, the "
                                                    "first 3 polytopes pulled in by 0.2 m) on the forest corridors: R is where the speed peaks and "
                                                    "three 1.5 m polytopes are too short to stop in, so about half of these safe problems are "
                                                    "infeasible for every factor — problems FASTER never poses"}}


def literal_leg(torch, make_pipe, run_step, fused, abi, B, inflight=8, steps=32):
    """The same step with the hand-off of SURVEY.md 8(d) taken to the letter (R may lie outside its shrunk corridor): the solved
    fraction of the safe problems, and the throughput of that variant measured like the timed region (`inflight` pipelines, `steps`
    steps; more of its safe problems are infeasible for every factor, i.e. run all ten trials)."""
    pipes = [make_pipe(-1.0) for _ in range(inflight)]
    for pp in pipes:
        run_step(pp, fused)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for k in range(steps):
        run_step(pipes[k % inflight], fused)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    frac = float(pipes[0].d_sres.cpu().numpy().view(abi.result_dtype)[:B]["solved"].mean())
    for pp in pipes:
        pp.ctx.close()
    return frac, B * steps / dt


def latency_leg(whole, faces, reps=200):
    """One replan at a time through the C++ class of the boundary is what FASTER itself does (SolverHip::genNewTraj, batch of 1:
    tests/cpp/test_solver_hip.cpp); here the same single-problem launch through the C ABI's host-pointer entry point — H2D of one
    problem record and its faces, one launch, D2H of one result — for `reps` different whole problems of the batch, one after the
    other.  The reference's replan period is 10 ms (faster.yaml:5)."""
    import numpy as np

    from faster_amd import capi

    ctx = capi.Context(0)
    ms, solved = [], 0
    try:
        for i in range(reps + 8):
            pr = whole[i: i + 1].copy()
            nf = int(pr["face_off"][0][pr["n_poly"][0]])
            fc = faces[int(pr["face_begin"][0]): int(pr["face_begin"][0]) + nf].copy()
            pr["face_begin"] = 0
            t = time.perf_counter()
            r = ctx.solve_batch(pr, fc)
            if i >= 8:
                ms.append(1e3 * (time.perf_counter() - t))
                solved += int(r["solved"][0])
    finally:
        ctx.close()
    ms = np.array(ms)
    return {"median_ms": float(np.median(ms)), "p95_ms": float(np.percentile(ms, 95)), "max_ms": float(ms.max()), "replans": int(len(ms)),
            "solved": solved, "replan_period_of_the_reference_ms": 10.0,
            "note": "fh_solve_batch with ONE whole problem (N=10, <=6 polytopes) per call, host pointers: copies, launch and synchronisation "
                    "included; the batch entry points exist for throughput, this is the latency a single SolverHip::genNewTraj() sees"}


def cpu_baseline(whole, faces, safe, sfaces, target_s):
    """The CPU oracle (oracle/faster_oracle.c, kind "port": Gurobi is absent) timed on a bounded sample of the SAME pairs:
    OpenMP over problems on the cores this process may use (affinity mask, cgroup quota), repeated until about `target_s`
    seconds of wall time have been spent.  Also the single-thread rate and the rate on 8 threads (how the port scales).
    Reported baseline only."""
    from oracle import oracle as orc

    orc.build()
    cores = max(1, min(host_cores(), orc.max_threads()))

    work = {}

    def run(k, threads):
        t = time.perf_counter()
        w = orc.solve_batch(whole[:k], faces, threads=threads)
        act = safe[:k][safe[:k]["n_seg"] > 0]
        it, nd, tr = float(w["qp_iters"].sum()), float(w["nodes"].sum()), float(w["trials"].sum())
        if len(act):
            r = orc.solve_batch(act, sfaces, threads=threads)
            it, nd, tr = it + float(r["qp_iters"].sum()), nd + float(r["nodes"].sum()), tr + float(r["trials"].sum())
        dt = time.perf_counter() - t
        work.update(mean_qp_iters_per_pair=it / k, mean_bnb_nodes_per_pair=nd / k, mean_trials_per_pair=tr / k)
        return dt

    k1 = min(512, len(whole))
    t1 = run(k1, 1)                       # single thread
    k8 = min(4096, len(whole))
    run(min(len(whole), 4096), cores)     # warm up the thread pool
    t8 = run(k8, min(8, cores))
    k = len(whole)
    passes, spent = 0, 0.0
    while spent < target_s and passes < 64:
        spent += run(k, cores)
        passes += 1
    return {"value": passes * k / spent, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": "%d passes over the %d pairs of rank 0 (whole + safe solves), CPU restatement oracle/faster_oracle.c with OpenMP "
                      "over problems on %d threads (os.sched_getaffinity / cgroup quota; os.cpu_count() = %d); NOT Gurobi (absent)"
                      % (passes, k, cores, os.cpu_count() or 0),
            "seconds": spent, "single_thread_value": k1 / t1, "single_thread_sample": "%d pairs" % k1,
            "eight_thread_value": k8 / t8, "scaling_vs_one_thread": (passes * k / spent) / (k1 / t1),
            # what the port DOES per pair (full jerk space, cold-started nodes, no box bound, no child bound): the GPU/CPU ratio mixes
            # algorithm and hardware — compare with config.mean_qp_iters_per_pair of the device
            **work}


if __name__ == "__main__":
    main()
