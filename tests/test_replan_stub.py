"""Next row N2: the ROS/PCL-free restatement of Faster::replan (faster_amd/host/replan_stub.hpp) drives the solver
through the SolverGurobi surface in a closed loop (tests/cpp/test_replan_stub.cpp): unknown space, whole + safe solves, plan splice,
factor-window adaptation.  On CPU the two C-ABI calls of SolverHip are answered by the oracle (test infrastructure only)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_replan_stub")


def build():
    from faster_amd import build as fb
    from oracle import oracle as orc

    fb.build_all()
    orc.build()
    src = os.path.join(ROOT, "tests", "cpp", "test_replan_stub.cpp")
    deps = [src, os.path.join(ROOT, "tests", "cpp", "oracle_solver.hpp"), fb.HOST_SO] + [os.path.join(ROOT, "faster_amd", "host", f) for f in
                                                                                         ("replan_stub.hpp", "corridor_frontend.hpp", "corridor_frontend.cpp", "solver_hip.hpp", "decomp_hip.hpp")]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++14", "-DWITH_ORACLE", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "faster_amd", "host"),
                               src, os.path.join(ROOT, "faster_amd", "host", "corridor_frontend.cpp"), "-o", EXE, "-L", os.path.join(ROOT, "faster_amd"),
                               "-lsolverhip", "-lfasterhip", "-ldl", "-fopenmp", "-Wl,-rpath," + os.path.join(ROOT, "faster_amd")])
    return EXE


def check(out):
    assert out["reached"] == 1, out
    assert out["dist_to_goal"] < 0.3
    assert out["committed"] >= 10
    # failures are mostly "no safe trajectory from R" (stage 3): the reference then keeps the committed plan, as here
    assert out["stages"][1] == 0 and out["stages"][4] == 0
    assert out["safe_needed"] >= 5                      # the 3 m sensing radius (< Ra) forces safe trajectories
    assert out["min_clearance"] >= 0.2 - 1e-3           # never closer to a tree than the drone radius used for the corridors
    # consecutive goals are 10 ms apart; at a splice the reference erases A itself and appends A + DC (appendToPlan,
    # faster.cpp:606-648), i.e. one 20 ms step: continuity means no jump beyond that
    assert out["max_jump"] <= 2.1 * 0.01 * out["max_speed_norm"] + 2e-3
    assert out["max_speed"] <= 1.5 * 1.1                # the reference bounds velocity at segment starts only (solverGurobi.cpp:393-405)


@pytest.mark.parametrize("seed", [1, 2])
def test_closed_loop_with_oracle_backed_solver(seed):
    exe = build()
    r = subprocess.run([exe, "oracle", os.path.join(ROOT, "oracle", "liboracle.so"), str(seed)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    check(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))


@pytest.mark.gpu
def test_closed_loop_on_gpu():
    exe = build()
    r = subprocess.run([exe, "gpu", "-", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    check(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))


@pytest.mark.gpu
def test_closed_loop_on_gpu_with_device_decomposition():
    """Same closed loop with the corridor decomposition on the device as well (DecompHip = fh_decompose_batch)."""
    exe = build()
    r = subprocess.run([exe, "gpu-decomp", "-", "1"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    check(json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]))
