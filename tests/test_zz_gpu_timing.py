"""Row a12 on the GPU: cooperative cancellation and the deadline (SolverGurobi::StopExecution / ResetToNormalState and the abort flag
that ends genNewTraj's loop, /root/reference/faster/src/solverGurobi.cpp:15-39, :445-446, :474, :643-646).

These are the only GPU tests whose outcome depends on how LONG something takes, so:
  * the tests are marked `timing`, which tests/conftest.py collects LAST (and the file name sorts last as well) — every parity test of every row runs before a timing test can stop `pytest -x`;
  * no test assumes that a particular problem is slow.  `long_search()` climbs a ladder of ever more expensive searches (more factor
    trials, no child bound, one wavefront per problem, the batch repeated) and MEASURES each rung with a deadline probe: a rung is used
    only if a launch of it was still running after `min_ms`, i.e. an un-stopped run takes at least ten times the delay after which the
    stopper fires.  A speed-up of the kernels moves the tests up the ladder; only if no rung is long enough do they skip (with the
    measured times) — they cannot fail for being fast.
What is asserted is what the reference's caller can observe: the status (interrupted, unsolved), that the call returns promptly after the
request, that the request stays raised until it is cleared, and that the next call after the reset solves normally."""
import os
import subprocess
import threading
import time

import numpy as np
import pytest
import torch  # noqa: F401  (before libfasterhip.so is loaded: one HIP runtime per process)

from faster_amd import abi, capi, corridor

pytestmark = [pytest.mark.gpu, pytest.mark.timing]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STOP_DELAY_S = 0.03          # the stopper fires this long after the launch was started
MIN_RUN_MS = 300.0           # an un-stopped launch must still be running after this (10 x the delay)


def hard_problems(n=64, seed=77, f_inc=0.5):
    """Dense N=15 / P=8 corridors pulled 0.76 m inwards: mostly infeasible for every factor of the window [1, 10] (19 trials at
    f_inc = 0.5).  Most are rejected at once; a few need thousands of branch-and-bound nodes per trial (exact enumeration without an
    incumbent)."""
    pr, faces, _ = corridor.whole_batch(n, seed=seed, n_seg=15, p_choices=(8,), f_inc=f_inc)
    faces = faces.copy()
    faces["b"] -= 0.76
    return pr, faces


# (factor increment, fh_sched.child_bound, fh_params.share, copies of the 64-problem batch): each rung costs more than the one before
LADDER = [(0.5, 1, 1, 1), (0.5, 0, 1, 1), (0.05, 0, 1, 1), (0.05, 0, 0, 1), (0.005, 0, 0, 4), (0.0025, 0, 0, 16)]


def _configure(c, child_bound, share, deadline_ms):
    par = abi.default_params()
    par["share"] = share
    par["deadline_ms"] = deadline_ms
    c.set_params(par)
    c.set_sched(child_bound=child_bound)


def long_search(c, launch, min_ms=MIN_RUN_MS):
    """The first rung of LADDER on which `launch(c, pr, faces)` is still running after min_ms — measured: the launch is given a deadline
    of min_ms and must report at least one problem FH_ST_INTERRUPTED.  Leaves the context configured for that rung with NO deadline;
    returns (pr, faces, rung, log).  Skips the calling test if even the last rung finishes sooner."""
    log = []
    for rung in LADDER:
        f_inc, child_bound, share, copies = rung
        pr, faces = hard_problems(n=64, f_inc=f_inc)
        if copies > 1:
            pr, faces = corridor.concat([(pr, faces)] * copies)
        _configure(c, child_bound, share, min_ms)
        t = time.perf_counter()
        res = launch(c, pr, faces)
        dur = 1e3 * (time.perf_counter() - t)
        hit = int((res["status"] == abi.FH_ST_INTERRUPTED).sum())
        log.append("rung %s: %.0f ms, %d interrupted by the %.0f ms probe" % (rung, dur, hit, min_ms))
        if hit >= 1:
            _configure(c, child_bound, share, 0.0)
            print("\n".join(log))
            return pr, faces, rung, log
    _configure(c, 1, 1, 0.0)
    pytest.skip("no search of the ladder runs for %.0f ms on this device: %s" % (min_ms, "; ".join(log)))


def n_trials(f_inc):
    f, k = 1.0, 0
    while f <= 10.0:       # genNewTraj's loop: for (double i = f_init; i <= f_final; i += f_inc), solverGurobi.cpp:445
        k += 1
        f += f_inc
    return k


def test_stop_request_from_another_thread():
    """a12: SolverGurobi::StopExecution() is meant to be called from another thread while m.optimize() runs (solverGurobi.cpp:15-39).
    fh_request_stop() raises a word in mapped host memory; the workgroups poll it between branch-and-bound nodes."""
    c = capi.Context(0)
    try:
        easy, efaces, _ = corridor.whole_batch(64, seed=5)
        c.solve_batch(easy, efaces)                       # (first launch: allocations)
        pr, faces, rung, _ = long_search(c, lambda c, p, f: c.solve_batch(p, f))
        out = {}

        def run():
            t = time.perf_counter()
            out["res"] = c.solve_batch(pr, faces)
            out["t_end"] = time.perf_counter()
            out["dur"] = out["t_end"] - t

        th = threading.Thread(target=run)
        th.start()
        time.sleep(STOP_DELAY_S)                          # the launch is under way (it takes >= MIN_RUN_MS un-stopped: measured above)
        t_stop = time.perf_counter()
        c.request_stop()
        th.join(timeout=60)
        assert not th.is_alive()
        latency = out["t_end"] - t_stop
        res = out["res"]
        hit = res["status"] == abi.FH_ST_INTERRUPTED
        print("rung %s: stop latency %.3f ms (launch had run %.0f ms), %d of %d interrupted" % (rung, 1e3 * latency, 1e3 * (out["dur"] - latency), hit.sum(), len(res)))
        assert hit.sum() >= 1 and np.all(res["solved"][hit] == 0)
        assert 0 <= latency < 0.02, latency               # measured ~1 ms: poll every few nodes + D2H of the results
        # the request stays raised: the next launch returns at once (a problem that costs nothing may still finish before its workgroup
        # sees the word; what had to be interrupted before is interrupted again) ...
        t = time.perf_counter()
        again = c.solve_batch(pr, faces)
        assert time.perf_counter() - t < 1e-3 * MIN_RUN_MS / 3
        ahit = again["status"] == abi.FH_ST_INTERRUPTED
        assert ahit.sum() >= 1 and np.all(again["solved"][ahit] == 0)
        # ... until it is cleared (ResetToNormalState): the next call solves normally
        c.clear_stop()
        _configure(c, 1, 1, 0.0)
        r = c.solve_batch(easy, efaces)
        assert r["solved"].sum() > 50 and not np.any(r["status"] == abi.FH_ST_INTERRUPTED)
    finally:
        c.clear_stop()
        c.close()


def test_deadline():
    """fh_params.deadline_ms: a wall-clock budget for a launch (the reference's replan period is 10 ms, faster.yaml:5)."""
    c = capi.Context(0)
    try:
        easy, efaces, _ = corridor.whole_batch(256, seed=6)
        c.solve_batch(easy, efaces)                       # (first launch: allocations)
        pr, faces, rung, _ = long_search(c, lambda c, p, f: c.solve_batch(p, f))
        allpr, allfaces = corridor.concat([(easy, efaces), (pr, faces)])
        _configure(c, rung[1], rung[2], 10.0)
        t = time.perf_counter()
        res = c.solve_batch(allpr, allfaces)
        dur = time.perf_counter() - t
        assert dur < 1e-3 * MIN_RUN_MS / 3, dur           # the 10 ms budget + copies, not the >= MIN_RUN_MS of the un-stopped search
        assert res["solved"][:256].sum() > 200            # the easy ones were done long before the deadline
        hit = res["status"][256:] == abi.FH_ST_INTERRUPTED
        assert hit.sum() >= 1 and np.all(res["solved"][256:][hit] == 0)
        _configure(c, 1, 1, 0.0)
        r = c.solve_batch(easy, efaces)
        assert not np.any(r["status"] == abi.FH_ST_INTERRUPTED)
    finally:
        c.close()


def test_speculative_search_ends_on_stop_and_deadline():
    """fh_solve_batch_speculative (what SolverHip::genNewTraj calls with setConcurrentFactors) with several factors in flight:
    FH_ST_INTERRUPTED is terminal — the search of a problem stops there, unsolved, instead of going on to later factor windows
    (genNewTraj's loop ends on the abort flag, solverGurobi.cpp:445-446, :643-646) — and fh_params.deadline_ms is ONE budget for the
    whole search."""
    c = capi.Context(0)
    try:
        easy, efaces, _ = corridor.whole_batch(64, seed=6)
        c.solve_batch_speculative(easy, efaces, 4)        # (first launch: allocations)
        pr, faces, rung, _ = long_search(c, lambda c, p, f: c.solve_batch_speculative(p, f, 4))
        _configure(c, rung[1], rung[2], 10.0)
        t = time.perf_counter()
        res = c.solve_batch_speculative(pr, faces, 4)     # the factors of a problem 4 at a time: several windows
        dur = time.perf_counter() - t
        assert dur < 1e-3 * MIN_RUN_MS / 3, dur           # one 10 ms budget (+ copies), not one per window
        hit = res["status"] == abi.FH_ST_INTERRUPTED
        assert hit.sum() >= 1 and np.all(res["solved"][hit] == 0) and np.all(res["factor"][hit] == 0)
        assert np.all(res["trials"][hit] <= n_trials(rung[0]))     # ended where they were interrupted (never beyond the window)
        _configure(c, rung[1], rung[2], 0.0)
        # the stop word: raised before the call, every search ends at the first window that holds an interrupted trial
        c.request_stop()
        t = time.perf_counter()
        res = c.solve_batch_speculative(pr, faces, 4)
        assert time.perf_counter() - t < 1e-3 * MIN_RUN_MS / 3
        hit = res["status"] == abi.FH_ST_INTERRUPTED
        # (a trial that costs nothing may still complete as infeasible before its workgroup sees the word and the search then moves on to
        # the next window; the first interrupted trial ends it)
        assert hit.sum() >= 1 and np.all(res["solved"][hit] == 0) and np.all(res["factor"][hit] == 0)
        c.clear_stop()
        _configure(c, 1, 1, 0.0)
        # and without either the speculative search equals the sequential one on the easy batch
        a, b = c.solve_batch_speculative(easy, efaces, 4), c.solve_batch(easy, efaces)
        for f in ("solved", "trials", "factor", "dt", "cost", "status", "assign"):
            assert np.array_equal(a[f], b[f]), f
    finally:
        c.clear_stop()
        c.close()


def test_solver_hip_stop_execution_from_another_thread(tmp_path):
    """The same through the C++ class: SolverHip::StopExecution() from another thread during genNewTraj() (tests/cpp/test_stop.cpp).
    The program finds its own long search: it makes the factor increment smaller (more trials of a corridor that no factor solves) until
    a genNewTraj() is still running when a watchdog stops it after MIN_RUN_MS; exit code 77 = no increment was long enough (skip)."""
    from faster_amd import build as fb

    fb.build_all()
    exe = os.path.join(ROOT, "tests", "cpp", "test_stop")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(fb.HOST_SO)):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "faster_amd", "host"), src,
                               "-o", exe, "-L", os.path.join(ROOT, "faster_amd"), "-lsolverhip", "-lfasterhip", "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    pr, faces = hard_problems(n=64)
    probe = capi.Context(0)
    hardest = int(np.argmax(probe.solve_batch(pr, faces)["nodes"]))   # the corridor whose 19 refuted trials cost most
    probe.close()
    p = pr[hardest]
    lines = ["15 0.01 5 5 8", " ".join(repr(float(v)) for v in p["x0"]), " ".join(repr(float(v)) for v in p["xf"][:3]), str(int(p["n_poly"]))]
    for k in range(int(p["n_poly"])):
        f0, f1 = p["face_begin"] + p["face_off"][k], p["face_begin"] + p["face_off"][k + 1]
        lines.append(str(f1 - f0))
        for f in range(f0, f1):
            lines.append("%r %r %r %r" % (float(faces["a"][f][0]), float(faces["a"][f][1]), float(faces["a"][f][2]), float(faces["b"][f])))
    sc = tmp_path / "hard.txt"
    sc.write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, str(sc), str(1e3 * STOP_DELAY_S), str(MIN_RUN_MS)], capture_output=True, text=True, timeout=300)
    print(r.stdout[-1200:])
    if r.returncode == 77:
        pytest.skip("no factor increment makes this search long enough: " + r.stdout[-600:])
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "STOP_OK" in r.stdout
