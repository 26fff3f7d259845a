"""Latency of single genNewTraj() calls through the host-pointer entry point (batch of 1) and small batches."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from faster_amd import abi, capi, corridor
ctx = capi.Context(0)
fx = json.load(open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fixture_corridor.json")))
cases = {"KA-1 whole N=10 P=3 (3 trials)": corridor.fixture_problem(fx, 10, (5, 3, 5), True, [0, 1, 2], fx["x0"], fx["xf"]),
         "yaml-default whole N=6 P=3": corridor.fixture_problem(fx, 6, (5, 5, 8), True, [0, 1, 2], fx["x0"], fx["xf"])}
w, f, _ = corridor.whole_batch(256, seed=3, p_choices=(2, 3, 4, 5, 6))
for name, (pr, faces) in cases.items():
    ctx.solve_batch(pr, faces)
    ts = []
    for _ in range(50):
        t = time.perf_counter(); r = ctx.solve_batch(pr, faces); ts.append(time.perf_counter() - t)
    print("%-34s median %.3f ms  min %.3f ms  kernel %.3f ms  (iters %d nodes %d)" % (name, 1e3 * np.median(ts), 1e3 * min(ts), ctx.last_kernel_ms(), r["qp_iters"][0], r["nodes"][0]))
    for width in (3, 10):  # N4: the same line search, `width` factors at a time
        ts = []
        for _ in range(50):
            t = time.perf_counter(); r2 = ctx.solve_batch_speculative(pr, faces, width); ts.append(time.perf_counter() - t)
        assert all(np.array_equal(r2[k], r[k]) for k in abi.result_dtype.names if k not in ("nodes", "qp_iters", "kflops"))
        print("%-34s   width %2d: median %.3f ms  min %.3f ms  (trials %d)" % ("", width, 1e3 * np.median(ts), 1e3 * min(ts), r2["trials"][0]))
for n in (1, 16, 256):
    ts = []
    for _ in range(20):
        t = time.perf_counter(); r = ctx.solve_batch(w[:n], f); ts.append(time.perf_counter() - t)
    print("synthetic C4-whole batch %4d: median %.3f ms (%.1f us/solve) kernel %.3f ms" % (n, 1e3 * np.median(ts), 1e6 * np.median(ts) / n, ctx.last_kernel_ms()))
    ts = []
    for _ in range(20):
        t = time.perf_counter(); r2 = ctx.solve_batch_speculative(w[:n], f, 10); ts.append(time.perf_counter() - t)
    print("   same, 10 factors at a time      : median %.3f ms (mean trials %.2f)" % (1e3 * np.median(ts), r2["trials"].mean()))
# PCIe-inclusive throughput of the host-pointer path at the bench batch size
w, f, _ = corridor.whole_batch(32768, seed=3, p_choices=(2, 3, 4, 5, 6))
ctx.solve_batch(w, f)
t = time.perf_counter(); ctx.solve_batch(w, f); dt = time.perf_counter() - t
print("host-pointer path, 32768 whole solves incl. H2D/D2H: %.1f ms => %.0f solves/s (kernel %.1f ms)" % (1e3 * dt, 32768 / dt, ctx.last_kernel_ms()))
