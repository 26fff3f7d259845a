"""The C++ host class `SolverHip` (faster_amd/host/solver_hip.hpp) keeps the SolverGurobi surface
(/root/reference/faster/include/solverGurobi.hpp:61-137) and is driven with the call sequence of
Faster (faster/src/faster.cpp:52-71, :406-427, :521-537, :582-588) by tests/cpp/test_solver_hip.cpp."""
import json
import os
import subprocess

import numpy as np
import pytest

from faster_amd import abi, corridor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_solver_hip")


def build_driver():
    from faster_amd import build as fb

    fb.build_all()
    src = os.path.join(ROOT, "tests", "cpp", "test_solver_hip.cpp")
    deps = [src, fb.HOST_SO, os.path.join(ROOT, "faster_amd", "host", "solver_hip.hpp")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O1", "-std=c++14", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "faster_amd", "host"),
                               src, "-o", EXE, "-L", os.path.join(ROOT, "faster_amd"), "-lsolverhip", "-lfasterhip",
                               "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    return EXE


def parse(stdout):
    """The class prints the reference's own console messages (e.g. StopExecution) on stdout: keep the JSON lines."""
    return json.loads("\n".join(l for l in stdout.splitlines() if l[:1] in ('{', '}', '"')))


def write_scenario(path, fx):
    P = fx["polytopes"]
    lines = ["10 6 0.01 5 3 5 1.0 1.0 20 20", " ".join(map(str, fx["x0"])), " ".join(map(str, fx["xf"][:3]))]

    def polys(idx):
        out = [str(len(idx))]
        for p in idx:
            out.append(str(len(P[p]["b"])))
            for a, b in zip(P[p]["A"], P[p]["b"]):
                out.append("%r %r %r %r" % (a[0], a[1], a[2], b))
        return out

    lines += polys([0, 1, 2])
    lines.append("13 11.5 3")          # M: second vertex of decomp_test_node/data/path3d.txt
    lines += polys([0])
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_host_class_compiles_and_links():
    """CPU-side: the SolverGurobi-shaped class and its Faster-style driver compile and link against the C ABI."""
    exe = build_driver()
    assert os.path.exists(exe)
    hdr = open(os.path.join(ROOT, "faster_amd", "host", "solver_hip.hpp")).read()
    for name in ("setN", "setX0", "setXf", "resetX", "setBounds", "genNewTraj", "getDTInitial", "setDC", "setPolytopes", "fillX",
                 "setForceFinalConstraint", "setWMax", "setMaxConstraints", "createVars", "setThreads", "setVerbose", "StopExecution",
                 "ResetToNormalState", "setMode", "setFactorInitialAndFinalAndIncrement", "X_temp_", "dt_", "trials_", "temporal_",
                 "runtime_ms_", "factor_that_worked_", "N_", "cb_"):
        assert name in hdr, name


def test_host_class_fails_loudly_without_gpu(tmp_path, fixture_corridor):
    """No device => genNewTraj() returns false (never throws, never falls back to a CPU path)."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = build_driver()
    sc = tmp_path / "scenario.txt"
    write_scenario(sc, fixture_corridor)
    r = subprocess.run([exe, str(sc)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    out = parse(r.stdout)
    assert out["whole"]["solved"] == 0 and out["safe"]["solved"] == 0
    assert "no HIP device" in r.stderr or "device error" in r.stderr


@pytest.mark.gpu
def test_host_class_matches_oracle(tmp_path, oracle, fixture_corridor):
    exe = build_driver()
    sc = tmp_path / "scenario.txt"
    write_scenario(sc, fixture_corridor)
    r = subprocess.run([exe, str(sc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = parse(r.stdout)
    assert out["cancelled"] == {"ret": 0, "trials": 0, "flag_after": 0}
    fx = fixture_corridor
    # whole: SURVEY.md App. B anchor KA-1
    pw, fw = corridor.fixture_problem(fx, 10, (5, 3, 5), True, [0, 1, 2], fx["x0"], fx["xf"])
    rw = oracle.solve_batch(pw, fw)[0]
    w = out["whole"]
    assert w["solved"] == 1 and w["trials"] == rw["trials"] == 3 and w["factor"] == rw["factor"] == 3.0
    assert w["dt"] == rw["dt"]
    assert w["cost"] == pytest.approx(23.869158339522752, rel=1e-9)
    Xw = oracle.sample(pw[0], rw)
    assert w["n"] == Xw.shape[0]
    np.testing.assert_allclose(w["first"][:3], Xw[0]["pos"], atol=1e-9)
    np.testing.assert_allclose(w["last"][:3], Xw[-1]["pos"], atol=1e-7)
    assert w["last"][3] == 0 and w["last"][5] == 0   # last sample: zero vel / jerk (solverGurobi.cpp:165-167)
    # safe from R = X_whole[n/2]
    R = Xw[Xw.shape[0] // 2]
    np.testing.assert_allclose(out["R"], np.concatenate([R["pos"], R["vel"], R["accel"]]), atol=1e-7)
    ps, fs = corridor.fixture_problem(fx, 6, (5, 3, 5), False, [0], out["R"], [13, 11.5, 3, 0, 0, 0, 0, 0, 0])
    rs = oracle.solve_batch(ps, fs)[0]
    s = out["safe"]
    assert s["solved"] == rs["solved"] and s["trials"] == rs["trials"]
    if rs["solved"]:
        assert s["factor"] == rs["factor"] and s["cost"] == pytest.approx(rs["cost"], rel=1e-7, abs=1e-9)
    # window update (faster.cpp:582-588): [max(3-20,1), 3+20] => same first feasible factor
    assert out["window"] == [1.0, 23.0]
    assert out["whole_again"]["solved"] == 1 and out["whole_again"]["factor"] == 3.0
    # concurrent factor search (SolverHip::setConcurrentFactors): bit-identical to the sequential whole solve
    c = out["whole_concurrent"]
    for k in ("solved", "trials", "factor", "dt", "cost", "n", "first", "last"):
        assert c[k] == w[k], k


def _build_decomp_driver():
    from faster_amd import build as fb

    fb.build_all()
    exe = os.path.join(ROOT, "tests", "cpp", "test_decomp_hip")
    src = os.path.join(ROOT, "tests", "cpp", "test_decomp_hip.cpp")
    deps = [src, fb.HOST_SO, os.path.join(ROOT, "faster_amd", "host", "decomp_hip.hpp")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "faster_amd", "host"), src,
                               "-o", exe, "-L", os.path.join(ROOT, "faster_amd"), "-lsolverhip", "-lfasterhip",
                               "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    return exe


def test_device_decomposition_fails_loudly_without_gpu():
    """DecompHip (the replan's cvxEllipsoidDecomp on the device) has no CPU fallback either."""
    import json

    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([_build_decomp_driver()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["polytopes"] == 0 and out["error"] == 1
    assert "DecompHip: device error" in r.stderr


@pytest.mark.gpu
def test_device_decomposition_through_the_host_class():
    import json

    r = subprocess.run([_build_decomp_driver()], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["polytopes"] == 2 and out["error"] == 0 and all(7 <= n <= 12 for n in out["rows"])


def _jps_scene(path, n_q=96, seed=9):
    from faster_amd import frontend

    cloud, cells, center, starts, goals = frontend.forest_queries(n_q, seed)
    with open(path, "w") as f:
        f.write("%d %d %d 0.2 0.3 0.0 3.0 %r %r %r %d %d\n" % (cells[0], cells[1], cells[2], float(center[0]), float(center[1]), float(center[2]), len(cloud), n_q))
        for p in cloud:
            f.write("%r %r %r\n" % (float(p[0]), float(p[1]), float(p[2])))
        for s, g in zip(starts, goals):
            f.write("%r %r %r %r %r %r\n" % tuple(float(v) for v in (*s, *g)))


def _build_jps_test():
    from faster_amd import build as fb

    fb.build_all()
    exe = os.path.join(ROOT, "tests", "cpp", "test_jps_hip")
    src = exe + ".cpp"
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(fb.HOST_SO)):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-fopenmp", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "faster_amd", "host"),
                               src, os.path.join(ROOT, "faster_amd", "host", "corridor_frontend.cpp"), "-o", exe, "-L", os.path.join(ROOT, "faster_amd"),
                               "-lsolverhip", "-lfasterhip", "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    return exe


def test_jps_hip_mirrors_jps_manager_and_fails_loudly_without_gpu(tmp_path):
    """JpsHip has JPS_Manager's path-search surface (jps_manager.hpp:40-59); without a device updateJPSMap reports failure."""
    import torch

    hdr = open(os.path.join(ROOT, "faster_amd", "host", "jps_hip.hpp")).read()
    for name in ("setNumCells", "setFactorJPS", "setResolution", "setInflationJPS", "setZGroundAndZMax", "updateJPSMap", "solveJPS3D"):
        assert name in hdr, name
    exe = _build_jps_test()
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sc = tmp_path / "scene.txt"
    _jps_scene(sc, n_q=4)
    r = subprocess.run([exe, str(sc)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "updateJPSMap failed" in r.stdout and "no HIP device" in r.stderr


@pytest.mark.gpu
def test_jps_hip_equals_host_search(tmp_path):
    """The JPS_Manager-shaped C++ class over fh_map_*: every path vertex for vertex as fhfront::plan_path_jps (its default: jump point
    search in jps3d's order, as JPS_Manager) and, switched to the A* of mode 0, as fhfront::plan_path (tests/cpp/test_jps_hip.cpp)."""
    exe = _build_jps_test()
    sc = tmp_path / "scene.txt"
    _jps_scene(sc, n_q=256)
    r = subprocess.run([exe, str(sc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "JPS_OK 256" in r.stdout and int(r.stdout.split()[-1]) > 240


def test_reference_types_build_round_trip(tmp_path, oracle, fixture_corridor):
    """INTEGRATION.md §1 prescribes -DFASTER_HIP_USE_REFERENCE_TYPES: SolverHip against FASTER's own `state`
    (faster/include/faster_types.hpp:79-165) and DecompUtil's LinearConstraint3D (decomp_geometry/polyhedron.h:115-185), included where
    they lie under /root/reference (Eigen through the test-only shim).  Compiles solver_hip.cpp that way and drives one replan's calls
    — setX0 / setXf / setPolytopes / genNewTraj / fillX — with the C-ABI calls answered by the CPU oracle: known answer KA-1.
    Skipped where /root/reference is absent (the GPU box)."""
    ref = os.environ.get("FASTER_REFERENCE", "/root/reference")
    inc_types = os.path.join(ref, "faster", "include")
    inc_decomp = os.path.join(ref, "thirdparty", "DecompROS", "DecompUtil", "include")
    if not (os.path.exists(os.path.join(inc_types, "faster_types.hpp")) and os.path.isdir(inc_decomp)):
        pytest.skip("the reference sources are not present")
    from oracle import oracle as orc

    exe = str(tmp_path / "test_reference_types")
    host = os.path.join(ROOT, "faster_amd", "host")
    subprocess.check_call(["g++", "-O1", "-std=c++14", "-w", "-DFASTER_HIP_USE_REFERENCE_TYPES",
                           "-I", os.path.join(ROOT, "oracle", "ref_frontend", "shim"), "-I", inc_types, "-I", inc_decomp,
                           "-I", os.path.join(ROOT, "include"), "-I", host, "-I", os.path.join(ROOT, "tests", "cpp"),
                           os.path.join(ROOT, "tests", "cpp", "test_reference_types.cpp"), os.path.join(host, "solver_hip.cpp"),
                           "-o", exe, "-L", os.path.join(ROOT, "faster_amd"), "-lfasterhip", "-ldl",
                           "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    fx = fixture_corridor
    sc = tmp_path / "fixture.txt"
    lines = [" ".join(repr(float(v)) for v in list(fx["x0"][:3]) + list(fx["xf"][:3])), str(len(fx["polytopes"]))]
    for p in fx["polytopes"]:
        lines.append(str(len(p["b"])))
        for a, b in zip(p["A"], p["b"]):
            lines.append("%r %r %r %r" % (float(a[0]), float(a[1]), float(a[2]), float(b)))
    sc.write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, orc.build(), str(sc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["geometry"] == 1
    assert out["solved"] == 1 and out["trials"] == 3 and out["factor"] == 3.0
    assert out["dt"] == 0.7348469495773315
    assert out["cost"] == pytest.approx(23.869158339522752, rel=1e-9)
    assert out["n"] == int(10 * out["dt"] / 0.01)
    np.testing.assert_allclose(out["first"], [p + 0.0 for p in fx["x0"][:3]], atol=1e-4)
    np.testing.assert_allclose(out["last"], fx["xf"][:3], atol=1e-6)
    assert out["last_vel_norm"] == 0.0
