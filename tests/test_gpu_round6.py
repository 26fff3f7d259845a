"""GPU tests added in round 6 (all through the C ABI).

Third-party parity AT THE BASELINE SIZES, ON GPU OUTPUT: Gurobi is absent, so `m.optimize()` (solverGurobi.cpp:566) cannot be
run here; what CAN be had from software the builder did not write is
  * HiGHS (scipy.optimize.milp) deciding the reference's mixed-integer constraint set — the unreduced 12 N coefficients, one binary per
    (segment, polytope), big-M indicator rows, oracle/py_model.py:milp_feasible — at the dt the GPU reports: feasible at
    factor_that_worked_ (solverGurobi.cpp:445-446, :580-581), infeasible at every earlier factor of the window, and infeasible at ALL ten
    factors for safe problems the GPU reports unsolved;
  * the global optimum over every one of the P^N assignments at N = 10 (59 049 pure QPs per problem through fh_problem.pin — the GPU
    solves them in a fraction of a second), with SciPy's SLSQP on the unreduced model confirming the winner's cost.
"""
import itertools
import multiprocessing
import os

import numpy as np
import pytest

from faster_amd import abi, capi, corridor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import torch  # noqa: F401  (torch first: one HIP runtime in the process)

    c = capi.Context(0)
    yield c
    c.close()


def _polys_of(p, faces):
    fb = int(p["face_begin"])
    return [(faces["a"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy(), faces["b"][fb + p["face_off"][q]: fb + p["face_off"][q + 1]].copy())
            for q in range(int(p["n_poly"]))]


def _run_highs(jobs):
    """jobs through oracle/py_model.py:milp_job (HiGHS: feasible? and, on a disagreement, whether 1e-6 of slack flips it) on all host
    cores.  The pool is SPAWNED, not forked: this process has the HIP runtime, its threads and (in a full run of the suite) a dozen
    contexts behind it — a forked child of that can hang before it runs a line of SciPy (seen: the suite stopped here for 15 minutes)."""
    from oracle import py_model

    workers = max(1, min(16, (os.cpu_count() or 2) - 1))
    try:
        with multiprocessing.get_context("spawn").Pool(workers) as pool:
            return pool.map(py_model.milp_job, jobs, chunksize=4)
    except Exception:
        return [py_model.milp_job(j) for j in jobs]


def _factor_jobs(tag, p, faces, r, base, n_seg, force, all_factors_if_unsolved=True):
    """The HiGHS questions one GPU result raises: feasible at its dt; infeasible at every earlier factor (or at every factor of the window)."""
    polys = _polys_of(p, faces)
    args = (n_seg, p["x0"].copy(), p["xf"].copy(), float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), force, polys)
    jobs = []
    f, k = float(p["f_init"]), 0
    while f <= float(p["f_final"]):
        dt = f * base
        k += 1
        if r["solved"] and k == int(r["trials"]):
            assert dt == r["dt"] and f == r["factor"], (tag, dt, r["dt"])
            jobs.append(((tag, k), True, args[0], dt) + args[1:])
            break
        if r["solved"] or all_factors_if_unsolved:
            jobs.append(((tag, k), False, args[0], dt) + args[1:])
        f = f + float(p["f_inc"])
    return jobs


def _judge(results, what, min_checks):
    bad = [(t, e, g) for t, e, g, m in results if g is not None and g != e and not m]
    marginal = [t for t, e, g, m in results if m]
    undecided = [t for t, e, g, m in results if g is None]
    n = len(results)
    print("%s: %d HiGHS verdicts (%d 'feasible', %d 'infeasible' expected), %d marginal (flip with 1e-6 of slack), %d undecided, %d disagree"
          % (what, n, sum(1 for r in results if r[1]), sum(1 for r in results if not r[1]), len(marginal), len(undecided), len(bad)))
    assert n >= min_checks, (what, n)
    assert not bad, (what, bad[:8])
    assert len(marginal) <= max(2, n // 100), (what, marginal[:8])
    assert len(undecided) <= max(2, n // 100), (what, undecided[:8])


def test_highs_confirms_flag_and_first_feasible_factor_of_c4_pairs(ctx):
    """BASELINE config C4 (N = 10, <= 6 polytopes whole / <= 3 safe): 512 pairs of the bench batch (seed 3) through the fused pair
    kernel.  For every whole result and every safe result HiGHS is asked about the reference's own mixed-integer constraint set
    (solverGurobi.cpp:180-291, :332-407, :499-524) at the GPU's dt: feasible at the factor the GPU reports, infeasible at every
    earlier factor of the window — and infeasible at all ten factors for (at least 64) safe problems the GPU reports unsolved.  This
    pins `solved` and factor_that_worked_ (:445-446, :580-581) against a solver the builder did not write."""
    from test_gpu_round3 import fused_pairs

    B = 512
    whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    wres, sres, safe, sfaces = fused_pairs(ctx, whole, faces, corridor.safe_templates(whole), 10, 0.05)
    base_w = np.maximum(ctx.dt_initial_batch(whole), 2 * whole["dc"])
    jobs = []
    for i in range(B):
        jobs += _factor_jobs(("whole", i), whole[i], faces, wres[i], float(base_w[i]), 10, True)
    res = _run_highs(jobs)
    _judge(res, "C4 whole problems", 512)
    assert wres["solved"].mean() > 0.99
    live = np.nonzero(safe["n_seg"] > 0)[0]
    base_s = np.maximum(ctx.dt_initial_batch(safe), 2 * safe["dc"])
    unsolved = [i for i in live if not sres[i]["solved"]]
    assert len(unsolved) >= 64, len(unsolved)
    jobs = []
    for i in live:
        if sres[i]["solved"] or i in unsolved[:64]:
            jobs += _factor_jobs(("safe", int(i)), safe[i], sfaces, sres[i], float(base_s[i]), 10, False)
    res = _run_highs(jobs)
    _judge(res, "C4 safe problems (incl. 64 unsolved: all ten factors)", 1000)
    all_ten = [t for t, e, g, m in res if t[0][1] in unsolved[:64]]
    assert len(all_ten) == 64 * 10


def test_highs_confirms_flag_and_first_feasible_factor_at_n15(ctx):
    """The same at BASELINE config C5's size: N = 15, <= 8 polytopes (96 synthetic corridors, 64 of them checked)."""
    pr, faces, _ = corridor.whole_batch(96, seed=615, n_seg=15, p_choices=(4, 5, 6, 7, 8))
    res = ctx.solve_batch(pr, faces)
    base = np.maximum(ctx.dt_initial_batch(pr), 2 * pr["dc"])
    jobs = []
    for i in range(64):
        jobs += _factor_jobs(("n15", i), pr[i], faces, res[i], float(base[i]), 15, True)
    _judge(_run_highs(jobs), "N = 15 whole problems", 64)
    assert res["solved"][:64].mean() > 0.8


_last_enumeration = (0, 0)


def _pinned_copies(p, n_seg, P, factor):
    """Every one of the P^N assignments of one problem as pinned copies solved at ONE factor (a pure QP each: BASELINE config 1's mechanism)."""
    combos = np.array(list(itertools.product(range(P), repeat=n_seg)), dtype=np.uint64)  # [P^N, N]
    w = np.zeros(len(combos), dtype=np.uint64)
    for t in range(n_seg):
        w |= (combos[:, t] + np.uint64(1)) << np.uint64(4 * t)
    out = np.repeat(p.reshape(1), len(combos))
    out["pin"][:, 0] = (w & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out["pin"][:, 1] = (w >> np.uint64(32)).astype(np.uint32)
    out["f_init"], out["f_final"], out["f_inc"] = factor, factor, 1.0
    return out, combos


def test_branch_and_bound_optimum_is_the_minimum_over_every_assignment_at_n10(ctx, seed_whole=610, seed_pairs=611):
    """Global optimality at N = 10 without trusting the tree search: for 24 whole problems with 3 polytopes (3^10 = 59 049 assignments
    each) and 16 safe problems of fused C4 pairs (<= 3 polytopes), EVERY assignment is solved as a pure QP through fh_problem.pin at the
    factor the branch and bound reported (the GPU does ~2 M QPs here).  The minimum over the feasible ones must be the branch and
    bound's cost (1e-9 relative) under the same or an equally cheap assignment, the EARLIER factors must have no feasible assignment
    at all, and SciPy's SLSQP on the reference's unreduced 12 N-coefficient model (oracle/py_model.py) must confirm the winner's cost.
    The reference has no way to fix its binaries (they are private, solverGurobi.cpp:208-230); with a Gurobi licence the same
    enumeration would be P = 1 corridors per assignment."""
    from oracle import py_model
    from test_gpu_round3 import fused_pairs

    whole, faces, _ = corridor.whole_batch(24, seed=seed_whole, n_seg=10, p_choices=(3,))
    wres = ctx.solve_batch(whole, faces)
    w2, f2, _ = corridor.whole_batch(96, seed=seed_pairs, n_seg=10, p_choices=(2, 3, 4, 5, 6))
    w2res, sres, safe, sfaces = fused_pairs(ctx, w2, f2, corridor.safe_templates(w2), 10, 0.05)
    pick = [j for j in np.nonzero((safe["n_seg"] > 0) & (sres["solved"] == 1) & (safe["n_poly"] >= 2))[0]][:16]
    assert len(pick) == 16
    cases = [("whole", whole[i], faces, wres[i], True) for i in range(len(whole)) if wres[i]["solved"]]
    cases += [("safe", safe[j], sfaces, sres[j], False) for j in pick]
    assert len(cases) >= 32
    total_qps = 0
    for kind, p, fcs, r, force in cases:
        P = int(p["n_poly"])
        # the winning factor: the minimum over all assignments
        copies, combos = _pinned_copies(p, 10, P, float(r["factor"]))
        got = ctx.solve_batch(copies, fcs)
        total_qps += len(copies)
        ok = got["solved"] == 1
        assert ok.any(), kind
        assert np.all(got["dt"][ok] == r["dt"])
        best = int(np.argmin(np.where(ok, got["cost"], np.inf)))
        assert got["cost"][best] == pytest.approx(r["cost"], rel=1e-9, abs=1e-12), (kind, got["cost"][best], r["cost"])
        assert r["cost"] <= got["cost"][ok].min() * (1 + 1e-9) + 1e-12
        if list(combos[best]) != [int(a) for a in r["assign"][:10]]:   # another assignment of the same cost (two polytopes hold the segment)
            mine = np.nonzero((combos == np.array(r["assign"][:10], dtype=np.uint64)).all(axis=1))[0]
            assert len(mine) == 1 and ok[mine[0]] and got["cost"][mine[0]] == pytest.approx(r["cost"], rel=1e-9, abs=1e-12)
        # every earlier factor of the window: no assignment at all is feasible (factor_that_worked_ is the FIRST feasible one)
        f = float(p["f_init"])
        while f < float(r["factor"]):
            earlier, _ = _pinned_copies(p, 10, P, f)
            e = ctx.solve_batch(earlier, fcs)
            total_qps += len(earlier)
            assert e["solved"].sum() == 0, (kind, f, int(e["solved"].sum()))
            f = f + float(p["f_inc"])
        # SciPy on the unreduced model confirms the winner
        s = py_model.solve_fixed(10, float(r["dt"]), p["x0"], p["xf"], float(p["v_max"]), float(p["a_max"]), float(p["j_max"]), force, _polys_of(p, fcs),
                                 [int(a) for a in combos[best]])
        assert s is not None and s[0] == pytest.approx(r["cost"], rel=1e-6, abs=1e-7), (kind, s and s[0], r["cost"])
    print("%d problems, %d pinned QPs on the GPU" % (len(cases), total_qps))
    assert total_qps > 1_500_000
    global _last_enumeration
    _last_enumeration = (len(cases), total_qps)  # (read by tests/tools/enumeration_sweep.py)


def test_lazy_pair_outputs_and_compact_results_give_the_same_bits():
    """fh_sched.pair_outputs = 0 (the library's default: the safe problem of a pair never leaves the chip unless it is shared between
    workgroups) and fh_sched.compact_results = 1 (only the coefficient rows the kernel is built for are written) against the complete
    outputs: every result field of every whole and safe problem bit for bit, the rows that are not written untouched, the caller's
    templates not written in their template fields — and with complete outputs the written safe problems equal the staged hand-off's
    (fh_pair_glue_device) record for record and row for row.  8192 C4 pairs (the bench batch's generator), sharing on: some safe
    problems ARE handed to other workgroups and staged from what write_safe_problem wrote."""
    import torch

    from tests.test_gpu_round3 import _dev

    B, N = 8192, 10
    whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    tmpl = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    RS = abi.result_dtype.itemsize

    def run(pair_outputs, compact, fill):
        c = capi.Context(0, pair_outputs=pair_outputs, compact_results=compact)
        c.set_pair_margin(0.05)
        d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(tmpl)
        d_sf = torch.zeros_like(d_faces)
        d_wr = torch.full((B * RS,), fill, dtype=torch.uint8, device="cuda:0")
        d_sr = torch.full((B * RS,), fill, dtype=torch.uint8, device="cuda:0")
        c.solve_pairs_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, 0.5, 0.2, 3, d_wr.data_ptr(), d_safe.data_ptr(), d_sf.data_ptr(), d_sr.data_ptr())
        c.sync()
        st = c.share_stats()
        out = (d_wr.cpu().numpy().view(abi.result_dtype), d_sr.cpu().numpy().view(abi.result_dtype), d_safe.cpu().numpy().view(abi.problem_dtype),
               d_sf.cpu().numpy().view(abi.face_dtype), st)
        c.close()
        return out

    w1, s1, safe1, sf1, st1 = run(True, False, 0xAB)
    w0, s0, safe0, sf0, st0 = run(False, True, 0xCD)
    fields = [f for f in abi.result_dtype.names if f not in ("nodes", "qp_iters", "kflops", "coeff")]
    for a, b, name in ((w1, w0, "whole"), (s1, s0, "safe")):
        for f in fields:
            assert np.array_equal(a[f], b[f]), (name, f)
        assert np.array_equal(a["coeff"][:, :N], b["coeff"][:, :N]), name
        # complete outputs: every word written (rows beyond the problem's segments are zero); compact: rows >= 10 keep the caller's bytes
        assert not a["coeff"][:, N:].any()
        assert (b["coeff"][:, N:].view(np.uint8) == 0xCD).all(), name
    assert w1["solved"].mean() > 0.99 and 0.5 < s1["solved"].mean() < 1.0
    # the template fields of the safe records are never written in either mode (n_seg = 0 marks a pair without a safe problem in the complete outputs only)
    live = w1["solved"] == 1
    for f in ("force_final_pos", "dc", "v_max", "a_max", "j_max", "f_init", "f_final", "f_inc", "xf", "pin"):
        assert np.array_equal(safe0[f], tmpl[f]) and np.array_equal(safe1[f], tmpl[f]), f
    assert np.array_equal(safe0["n_seg"], tmpl["n_seg"]) and np.array_equal(safe1["n_seg"][live], tmpl["n_seg"][live])
    # complete outputs = the staged hand-off's, record for record and row for row
    c = capi.Context(0)
    c.set_pair_margin(0.05)
    d_whole, d_faces, d_safe, d_wr = _dev(whole), _dev(faces), _dev(tmpl), _dev(w1)
    d_sf = torch.zeros_like(d_faces)
    c.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
    c.sync()
    safe_ref, sf_ref = d_safe.cpu().numpy().view(abi.problem_dtype), d_sf.cpu().numpy().view(abi.face_dtype)
    c.close()
    for f in abi.problem_dtype.names:
        assert np.array_equal(safe1[f], safe_ref[f]), f
    for i in np.flatnonzero(live):
        f0 = int(safe_ref["face_begin"][i]); n = int(safe_ref["face_off"][i][safe_ref["n_poly"][i]])
        assert np.array_equal(sf1["a"][f0:f0 + n], sf_ref["a"][f0:f0 + n]) and np.array_equal(sf1["b"][f0:f0 + n], sf_ref["b"][f0:f0 + n]), i
    # lazy outputs: what WAS written (pairs whose safe problem was shared) is the same record; everything else is the template
    written = np.flatnonzero((safe0["x0"] != tmpl["x0"]).any(axis=1))
    for i in written:
        for f in abi.problem_dtype.names:
            assert np.array_equal(safe0[f][i], safe_ref[f][i]), (i, f)
    print("lazy outputs: %d of %d safe records written (shared safe problems); donations %d / %d" % (len(written), B, st0["donated"], st1["donated"]))


def test_pairs_with_unusable_whole_problems_first_in_line():
    """A pair whose whole problem is unusable (bad input) or has no solution never stages a corridor — and the hand-off of the fused
    kernel runs on unconditionally and decides at its end.  Such problems as the FIRST units of their workgroups (nothing at all in LDS
    yet: launch order off, so unit = ticket, and the first tickets are dealt to the workgroups one chunk each) must give what the three
    launches give, with other launches of other contexts in flight (workgroups that start late take given tickets before they draw any)."""
    import torch

    from tests.test_gpu_round3 import _dev

    B, N = 16384, 10
    whole, faces, _ = corridor.whole_batch(B, seed=77, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    whole = whole.copy()
    whole["n_seg"][0:4096:4] = 0            # bad input
    whole["x0"][1:4096:4, 4] = np.nan       # bad input (found where x0 is staged)
    whole["f_final"][2:4096:4] = 1.0        # mostly without a solution
    whole["face_off"][3:4096:16, 1] = -5    # bad corridor layout
    tmpl = corridor.safe_templates(whole)
    mf = int(whole["face_off"][np.arange(B), np.clip(whole["n_poly"], 0, 8)].max())
    RS = abi.result_dtype.itemsize

    def staged(c):
        d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(tmpl)
        d_sf = torch.zeros_like(d_faces)
        d_wr, d_sr = torch.zeros(B * RS, dtype=torch.uint8, device="cuda:0"), torch.zeros(B * RS, dtype=torch.uint8, device="cuda:0")
        c.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, d_wr.data_ptr())
        c.pair_glue_device(d_whole.data_ptr(), d_wr.data_ptr(), d_faces.data_ptr(), B, 0.5, 0.2, 3, d_safe.data_ptr(), d_sf.data_ptr())
        c.solve_batch_device(d_safe.data_ptr(), d_sf.data_ptr(), B, N, mf, d_sr.data_ptr())
        c.sync()
        return d_wr.cpu().numpy().view(abi.result_dtype).copy(), d_sr.cpu().numpy().view(abi.result_dtype).copy()

    ref = capi.Context(0)
    ref.set_pair_margin(0.05)
    ref.set_sched(launch_order=0)
    wref, sref = staged(ref)
    ref.close()
    assert (wref["status"][0:4096:4] == abi.FH_ST_BAD_INPUT).all() and (wref["status"][1:4096:4] == abi.FH_ST_BAD_INPUT).all()
    lanes = []
    for k in range(4):  # four contexts, their fused launches in flight together, twice each
        c = capi.Context(0, pair_outputs=False, compact_results=True)
        s = torch.cuda.Stream()
        c.set_stream(s.cuda_stream)
        c.set_pair_margin(0.05)
        c.set_sched(launch_order=0)
        bufs = [_dev(whole), _dev(faces), _dev(tmpl)]
        bufs += [torch.zeros_like(bufs[1]), torch.zeros(B * RS, dtype=torch.uint8, device="cuda:0"), torch.zeros(B * RS, dtype=torch.uint8, device="cuda:0")]
        lanes.append((c, s, bufs))
    torch.cuda.synchronize()
    for rep in range(2):
        for c, s, (dw, df, ds, dsf, dwr, dsr) in lanes:
            c.solve_pairs_device(dw.data_ptr(), df.data_ptr(), B, N, mf, 0.5, 0.2, 3, dwr.data_ptr(), ds.data_ptr(), dsf.data_ptr(), dsr.data_ptr())
    torch.cuda.synchronize()
    fields = [f for f in abi.result_dtype.names if f not in ("nodes", "qp_iters", "kflops", "coeff")]
    for c, s, (dw, df, ds, dsf, dwr, dsr) in lanes:
        w, sr = dwr.cpu().numpy().view(abi.result_dtype), dsr.cpu().numpy().view(abi.result_dtype)
        for f in fields:
            assert np.array_equal(w[f], wref[f]), ("whole", f)
            assert np.array_equal(sr[f], sref[f]), ("safe", f)
        assert np.array_equal(w["coeff"][:, :N], wref["coeff"][:, :N]) and np.array_equal(sr["coeff"][:, :N], sref["coeff"][:, :N])
        assert c.share_stats()["error"] == 0
        c.close()


def test_a_launch_looks_around_less_often_while_other_launches_are_in_flight():
    """fh_sched.look_every = 0 (default): a solve launch that is issued while ANOTHER context of the process has a solve launch in flight on the
    device reads its control words every 16th node of a tree, a launch that has the device to itself every 8th (fh_last_launch reports
    the period; a fixed value is taken as it is, an invalid one refused).  The results are the same bit for bit."""
    import torch

    B, N = 16384, 10
    whole, faces, _ = corridor.whole_batch(B, seed=91, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    mf = int(whole["face_off"][np.arange(B), whole["n_poly"]].max())
    to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")
    d_whole, d_faces = to_dev(whole), to_dev(faces)
    a, b = capi.Context(0), capi.Context(0)
    try:
        outs = [torch.zeros(B * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0") for _ in range(3)]
        a.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, outs[0].data_ptr())
        a.sync()
        assert a.last_launch()[0]["look_every"] == 8
        alone = outs[0].cpu().numpy().view(abi.result_dtype).copy()
        # launches back to back on two contexts (two streams): the one on b is issued while a's are running (four of them queued on a's
        # stream — milliseconds of work — so that a stall of this host thread between the calls cannot let them finish first)
        a.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, outs[1].data_ptr())
        assert a.last_launch()[0]["look_every"] == 8
        for _ in range(3):
            a.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, outs[1].data_ptr())
        b.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, outs[2].data_ptr())
        a.sync(); b.sync()
        assert b.last_launch()[0]["look_every"] == 16
        for o in outs[1:]:
            got = o.cpu().numpy().view(abi.result_dtype)
            for f in ("solved", "trials", "status", "factor", "dt", "cost", "coeff", "assign"):
                assert np.array_equal(got[f], alone[f]), f
        b.set_sched(look_every=32)
        b.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), B, N, mf, outs[2].data_ptr())
        b.sync()
        assert b.last_launch()[0]["look_every"] == 32
        with pytest.raises(Exception):
            b.set_sched(look_every=12)
    finally:
        a.close(); b.close()
