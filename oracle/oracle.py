"""ctypes wrapper of the CPU oracle (oracle/faster_oracle.c).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see the header of faster_oracle.c): the reference arithmetic is Gurobi's, which is
absent; nothing here is Gurobi output.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

from faster_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "faster_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "fasterhip.h")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            so = build()
        L = ctypes.CDLL(so)
        L.orc_dt_initial.restype = ctypes.c_double
        L.orc_dt_initial.argtypes = [ctypes.c_void_p]
        L.orc_solve_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_solve_batch_mt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_solve_fixed.argtypes = [ctypes.c_void_p] * 5
        L.orc_miqp_dt.restype = ctypes.c_int
        L.orc_miqp_dt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        L.orc_bruteforce_dt.restype = ctypes.c_long
        L.orc_bruteforce_dt.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_double, ctypes.c_void_p]
        L.orc_sample.restype = ctypes.c_int
        L.orc_sample.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_dt_initial_batch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_control_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.orc_sizeof_problem.restype = ctypes.c_size_t
        L.orc_sizeof_result.restype = ctypes.c_size_t
        _LIB = L
    return _LIB


def _params(params):
    return abi.default_params() if params is None else params


def dt_initial(problem):
    pr = np.ascontiguousarray(problem).reshape(1)
    return lib().orc_dt_initial(abi.ptr(pr))


def dt_initial_batch(problems):
    pr = np.ascontiguousarray(problems)
    dt = np.zeros(pr.shape[0], dtype=np.float64)
    lib().orc_dt_initial_batch(abi.ptr(pr), pr.shape[0], abi.ptr(dt))
    return dt


def solve_batch(problems, faces, params=None, threads=None):
    """Oracle for fh_solve_batch: one genNewTraj() per problem."""
    problems = np.ascontiguousarray(problems)
    faces = np.ascontiguousarray(faces)
    res = np.zeros(problems.shape[0], dtype=abi.result_dtype)
    par = _params(params)
    # explicit thread count (an OMP_NUM_THREADS change after libgomp is loaded has no effect); None/0 = all cores
    lib().orc_solve_batch_mt(abi.ptr(problems), abi.ptr(faces), abi.ptr(par.reshape(1)), problems.shape[0], abi.ptr(res),
                             int(threads or 0))
    return res


def max_threads():
    return lib().orc_max_threads()


def solve_fixed(problem, faces, assign, params=None):
    """genNewTraj() with the binaries pinned (assign[t] >= 0) or free (-1)."""
    pr = np.ascontiguousarray(problem).reshape(1)
    a = np.full(abi.FH_MAX_SEG, -1, dtype=np.int8)
    a[: len(assign)] = assign
    res = np.zeros(1, dtype=abi.result_dtype)
    par = _params(params)
    lib().orc_solve_fixed(abi.ptr(pr), abi.ptr(np.ascontiguousarray(faces)), abi.ptr(par.reshape(1)), abi.ptr(a), abi.ptr(res))
    return res[0]


def miqp_dt(problem, faces, dt, assign=None, params=None):
    """One callOptimizer() at a given dt; assign pins segments (None = all free)."""
    pr = np.ascontiguousarray(problem).reshape(1)
    res = np.zeros(1, dtype=abi.result_dtype)
    par = _params(params)
    if assign is None:
        aptr = None
    else:
        a = np.full(abi.FH_MAX_SEG, -1, dtype=np.int8)
        a[: len(assign)] = assign
        aptr = abi.ptr(a)
    st = lib().orc_miqp_dt(abi.ptr(pr), abi.ptr(np.ascontiguousarray(faces)), abi.ptr(par.reshape(1)), dt, aptr, abi.ptr(res))
    return st, res[0]


def bruteforce_dt(problem, faces, dt, params=None):
    pr = np.ascontiguousarray(problem).reshape(1)
    res = np.zeros(1, dtype=abi.result_dtype)
    par = _params(params)
    nfeas = lib().orc_bruteforce_dt(abi.ptr(pr), abi.ptr(np.ascontiguousarray(faces)), abi.ptr(par.reshape(1)), dt, abi.ptr(res))
    return nfeas, res[0]


def control_points(results, n_seg):
    """orc_control_points: the Bezier control points of every result by the oracle's jerk-space route -> [n][n_seg][4][3]."""
    rs = np.ascontiguousarray(results)
    cp = np.zeros((rs.shape[0], int(n_seg), 4, 3), dtype=np.float64)
    lib().orc_control_points(abi.ptr(rs), rs.shape[0], int(n_seg), abi.ptr(cp))
    return cp


def sample(problem, result, max_samples=None):
    """resetX()+fillX(): returns the state array (count, ) of abi.state_dtype."""
    pr = np.ascontiguousarray(problem).reshape(1)
    rs = np.ascontiguousarray(result).reshape(1)
    if not rs[0]["solved"]:
        return np.zeros(0, dtype=abi.state_dtype)
    n = max(2, int(int(pr[0]["n_seg"]) * float(rs[0]["dt"]) / float(pr[0]["dc"])))
    cap = n if max_samples is None else max_samples
    out = np.zeros(cap, dtype=abi.state_dtype)
    cnt = lib().orc_sample(abi.ptr(pr), abi.ptr(rs), cap, abi.ptr(out))
    assert cnt == n, (cnt, n)
    return out[: min(cnt, cap)]
