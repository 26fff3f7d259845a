"""GPU parity: the HIP path (through the C ABI, faster_amd/libfasterhip.so) against the CPU oracle.

Bars: feasibility flag, trials_ and factor_that_worked_ exact; cost 1e-7 relative
(north-star bar: 1e-4); polynomial coefficients 1e-6 absolute.  Parity is UNPINNED with respect to Gurobi
(absent); the oracle itself is pinned in tests/test_oracle_*.py.
"""
import numpy as np
import pytest

from faster_amd import abi, corridor

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    # PyTorch-ROCm bundles its own HIP runtime: when a process uses both, torch must be imported BEFORE
    # libfasterhip.so is loaded so that both share one runtime.
    import torch  # noqa: F401

    from faster_amd import capi

    c = capi.Context(0)
    yield c
    c.close()


def compare(got, ref, n_seg=None, cost_rtol=1e-7, coeff_atol=1e-6):
    assert np.array_equal(got["status"] == abi.FH_ST_BAD_INPUT, ref["status"] == abi.FH_ST_BAD_INPUT)
    assert np.array_equal(got["solved"], ref["solved"]), np.nonzero(got["solved"] != ref["solved"])
    assert np.array_equal(got["trials"], ref["trials"])
    assert np.array_equal(got["factor"], ref["factor"])
    assert np.array_equal(got["dt"], ref["dt"])  # same float-cast time allocation, bit for bit
    assert np.array_equal(got["status"], ref["status"])
    ok = ref["solved"] == 1
    np.testing.assert_allclose(got["cost"][ok], ref["cost"][ok], rtol=cost_rtol, atol=1e-9)
    np.testing.assert_allclose(got["coeff"][ok], ref["coeff"][ok], rtol=0, atol=coeff_atol)
    return ok


def check_assignment_valid(pr, faces, res, tol=1e-6):
    """Every segment's 4 Bezier control points lie in the polytope the result assigns it to."""
    for i in np.nonzero(res["solved"] == 1)[0]:
        p, r = pr[i], res[i]
        h, N = r["dt"], int(p["n_seg"])
        for t in range(N):
            if p["n_poly"] == 0:
                assert r["assign"][t] == -1
                continue
            a, b, c, d = (r["coeff"][t][3 * k: 3 * k + 3] for k in range(4))
            cps = [d, d + c * h / 3, d + 2 * c * h / 3 + b * h * h / 3, a * h**3 + b * h**2 + c * h + d]
            q = int(r["assign"][t])
            assert 0 <= q < p["n_poly"]
            f0, f1 = p["face_begin"] + p["face_off"][q], p["face_begin"] + p["face_off"][q + 1]
            for cp in cps:
                assert np.max(faces["a"][f0:f1] @ cp - faces["b"][f0:f1]) <= tol


def test_known_answers_on_gpu(ctx, oracle, fixture_corridor, known_answers):
    batches = []
    for name in ("KA-1", "KA-2", "KA-3", "KA-4"):
        c = known_answers["cases"][name]
        batches.append(corridor.fixture_problem(fixture_corridor, c["N"], c["vaj"], c["force_final"], c["polys"], c["x0"], c["xf"]))
    pr, faces = corridor.concat(batches)
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)
    assert got["cost"][0] == pytest.approx(23.869158339522752, rel=1e-9)
    assert list(got["assign"][0][:10]) == [0, 0, 0, 0, 0, 0, 2, 2, 2, 2]
    assert got["cost"][2] == pytest.approx(22.019951865206, rel=1e-9)


@pytest.mark.parametrize("name,make", [
    ("C2 safe pure-QP N=6 P=1", lambda: corridor.safe_batch(1024, seed=1)),
    ("C3 whole N=10 P<=4", lambda: corridor.whole_batch(1024, seed=2)),
    ("C4 whole N=10 P<=6", lambda: corridor.whole_batch(512, seed=3, p_choices=(2, 3, 4, 5, 6))),
    ("C5 whole N=15 P<=8", lambda: corridor.whole_batch(128, seed=5, n_seg=15, p_choices=(4, 5, 6, 7, 8))),
    ("safe N=10 P<=3 multi-polytope", lambda: corridor.safe_batch(256, seed=7, n_seg=10, p_choices=(1, 2, 3))),
])
def test_parity_synthetic(ctx, oracle, name, make):
    pr, faces, _ = make()
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    ok = compare(got, ref)
    assert ok.mean() > 0.5, "generator should give mostly feasible problems"
    check_assignment_valid(pr, faces, got)


def test_edge_cases(ctx, oracle):
    pr, faces, _ = corridor.whole_batch(16, seed=21)
    pr = pr.copy()
    pr["n_poly"][0] = 0                      # no corridor at all
    pr["x0"][1, 3] = 7.0                     # initial speed above v_max: infeasible for every factor
    pr["n_seg"][2] = 0                       # bad input
    pr["n_seg"][3] = abi.FH_MAX_SEG + 1      # bad input
    pr["f_final"][4] = 0.5                   # empty factor window: zero trials
    pr["xf"][5, 0:3] = pr["x0"][5, 0:3]      # goal == start
    pr["f_inc"][6] = 0.0                     # bad input
    pr["x0"][7, 0] = np.nan                  # bad input
    pr["n_seg"][8] = 1
    pr["n_seg"][9] = 3
    pr["f_init"][10], pr["f_final"][10], pr["f_inc"][10] = 1.5, 4.0, 0.25   # non-integer window, accumulated in double
    got = ctx.solve_batch(pr, faces)
    ref = oracle.solve_batch(pr, faces)
    compare(got, ref)
    assert got["status"][2] == abi.FH_ST_BAD_INPUT and got["status"][6] == abi.FH_ST_BAD_INPUT
    assert got["trials"][4] == 0 and got["solved"][4] == 0
    assert got["solved"][1] == 0


def test_empty_batch(ctx):
    pr = abi.make_problems(0)
    faces = np.zeros(0, dtype=abi.face_dtype)
    assert ctx.solve_batch(pr, faces).shape == (0,)


def test_sampling_parity(ctx, oracle):
    pr, faces, _ = corridor.whole_batch(64, seed=31)
    res = ctx.solve_batch(pr, faces)
    cap = 4096
    states, counts = ctx.sample_batch(pr, res, cap)
    for i in range(len(pr)):
        ref = oracle.sample(pr[i], res[i])
        assert counts[i] == ref.shape[0]
        m = min(cap, ref.shape[0])
        for fld in ("pos", "vel", "accel", "jerk"):
            np.testing.assert_allclose(states[i, :m][fld], ref[:m][fld], rtol=0, atol=1e-11)
        if m:
            assert np.all(states[i, m - 1]["vel"] == 0) and np.all(states[i, m - 1]["jerk"] == 0)


def _dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to("cuda:0")


def test_pair_pipeline_device_pointers(ctx, oracle):
    """C4 pipeline with device-resident buffers: whole solve -> fh_pair_glue_device -> safe solve, against the
    oracle driven through the host restatement of the hand-off (oracle/pair_glue.py)."""
    import torch

    from oracle import pair_glue

    n, N = 384, 10
    whole, faces, _ = corridor.whole_batch(n, seed=3, n_seg=N, p_choices=(2, 3, 4, 5, 6))
    tmpl = corridor.safe_templates(whole)
    max_faces = int(whole["face_off"][np.arange(n), whole["n_poly"]].max())
    d_whole, d_faces, d_safe = _dev(whole), _dev(faces), _dev(tmpl)
    d_sfaces = torch.zeros_like(d_faces)
    d_wres = torch.zeros(n * abi.result_dtype.itemsize, dtype=torch.uint8, device="cuda:0")
    d_sres = torch.zeros_like(d_wres)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    ctx.timing_reset()
    ctx.solve_batch_device(d_whole.data_ptr(), d_faces.data_ptr(), n, N, max_faces, d_wres.data_ptr())
    ctx.pair_glue_device(d_whole.data_ptr(), d_wres.data_ptr(), d_faces.data_ptr(), n, 0.5, 0.2, 3, d_safe.data_ptr(), d_sfaces.data_ptr())
    ctx.solve_batch_device(d_safe.data_ptr(), d_sfaces.data_ptr(), n, N, max_faces, d_sres.data_ptr())
    ctx.sync()
    assert len(ctx.timing_read()) == 2 and ctx.last_kernel_ms() > 0
    ctx.set_stream(0)
    wres = d_wres.cpu().numpy().view(abi.result_dtype)
    sres = d_sres.cpu().numpy().view(abi.result_dtype)
    safe = d_safe.cpu().numpy().view(abi.problem_dtype)
    sfaces = d_sfaces.cpu().numpy().view(abi.face_dtype)

    wref = oracle.solve_batch(whole, faces)
    compare(wres, wref)
    # hand-off: same R, same corridor selection (inputs: the GPU's whole results, so that rounding cannot fork it)
    safe_ref, sfaces_ref = pair_glue.glue(whole, wres, faces, tmpl, 0.5, 0.2, 3)
    assert np.array_equal(safe["n_seg"], safe_ref["n_seg"])
    assert np.array_equal(safe["n_poly"], safe_ref["n_poly"])
    assert np.array_equal(safe["face_off"], safe_ref["face_off"])
    np.testing.assert_allclose(safe["x0"], safe_ref["x0"], rtol=0, atol=1e-11)
    live = safe["n_seg"] > 0
    for i in np.nonzero(live)[0]:
        f0 = safe["face_begin"][i]
        f1 = f0 + safe["face_off"][i][safe["n_poly"][i]]
        np.testing.assert_allclose(sfaces["a"][f0:f1], sfaces_ref["a"][f0:f1], rtol=0, atol=0)
        np.testing.assert_allclose(sfaces["b"][f0:f1], sfaces_ref["b"][f0:f1], rtol=0, atol=1e-14)
    sref = oracle.solve_batch(safe, sfaces)
    compare(sres, sref)
    check_assignment_valid(safe, sfaces, sres)
    assert 0.3 < sres["solved"].mean() <= 1.0
