"""numpy / ctypes mirror of the C ABI structs declared in include/fasterhip.h.

The layouts are checked against `sizeof` exported by the shared libraries in tests/test_abi.py.
Field meaning: see include/fasterhip.h (each field cites the SolverGurobi member it replaces).
"""
import ctypes

import numpy as np

FH_MAX_SEG = 16
FH_MAX_POLY = 8
FH_MAX_FACES = 256
FH_MAX_FACES_POLY = 64

FH_ST_OPTIMAL, FH_ST_INFEASIBLE, FH_ST_NODE_LIMIT, FH_ST_ITER_LIMIT, FH_ST_BAD_INPUT, FH_ST_INTERRUPTED = range(6)

problem_dtype = np.dtype(
    [
        ("n_seg", "<i4"),
        ("n_poly", "<i4"),
        ("force_final_pos", "<i4"),
        ("face_begin", "<i4"),
        ("face_off", "<i4", (FH_MAX_POLY + 1,)),
        ("pin", "<u4", (2,)),
        ("reserved", "<i4"),
        ("dc", "<f8"),
        ("v_max", "<f8"),
        ("a_max", "<f8"),
        ("j_max", "<f8"),
        ("f_init", "<f8"),
        ("f_final", "<f8"),
        ("f_inc", "<f8"),
        ("x0", "<f8", (9,)),
        ("xf", "<f8", (9,)),
    ],
    align=True,
)

face_dtype = np.dtype([("a", "<f8", (3,)), ("b", "<f8")], align=True)

result_dtype = np.dtype(
    [
        ("solved", "<i4"),
        ("trials", "<i4"),
        ("status", "<i4"),
        ("nodes", "<i4"),
        ("qp_iters", "<i4"),
        ("kflops", "<i4"),
        ("factor", "<f8"),
        ("dt", "<f8"),
        ("cost", "<f8"),
        ("coeff", "<f8", (FH_MAX_SEG, 12)),
        ("assign", "i1", (FH_MAX_SEG,)),
    ],
    align=True,
)

state_dtype = np.dtype([("pos", "<f8", (3,)), ("vel", "<f8", (3,)), ("accel", "<f8", (3,)), ("jerk", "<f8", (3,))], align=True)

params_dtype = np.dtype([("feas_tol", "<f8"), ("dep_tol", "<f8"), ("max_nodes", "<i4"), ("max_iters", "<i4"), ("max_work", "<i4"),
                         ("share", "<i4"), ("mip_gap", "<f8"), ("deadline_ms", "<f8")], align=True)
share_stats_dtype = np.dtype([(k, "<u4") for k in ("donated", "stolen", "queue_full", "records_full", "records_used", "error",
                                                   "interrupted", "workgroups")], align=True)

assert problem_dtype.itemsize == 264, problem_dtype.itemsize
assert face_dtype.itemsize == 32
assert result_dtype.itemsize == 1600, result_dtype.itemsize
assert state_dtype.itemsize == 96
assert params_dtype.itemsize == 48


sched_dtype = np.dtype([("launch_order", "<i4"), ("publish_factor", "<i4"), ("backlog", "<i4"), ("waiting_workgroups", "<i4"),
                        ("min_nodes", "<i4"), ("cloud_blocks", "<i4"), ("workgroups_per_cu", "<i4"), ("no_child_bound", "<i4"), ("compact_results", "<i4"), ("pair_outputs", "<i4"),
                        ("look_every", "<i4"), ("struct_size", "<i4")])
assert sched_dtype.itemsize == 48
FH_ABI_VERSION = 8   # include/fasterhip.h: the layout generation of its structs (checked against fh_abi_version() when the library is loaded)

launch_info_dtype = np.dtype([("n_seg", "<i4"), ("pairs", "<i4"), ("waves_per_simd", "<i4"), ("grid", "<i4"), ("workgroups_per_cu", "<i4"),
                              ("lds_bytes", "<i4"), ("unknown_space", "<i4"), ("look_every", "<i4")])


voxel_grid_dtype = np.dtype([("origin", "<f8", (3,)), ("res", "<f8"), ("dims", "<i4", (3,)), ("reserved", "<i4")])
assert voxel_grid_dtype.itemsize == 48


pair_rule_dtype = np.dtype([("mode", "<i4"), ("reserved", "<i4"), ("r_known", "<f8"), ("drone_radius", "<f8"), ("delta_h", "<f8"),
                            ("delta_a", "<f8")])
assert pair_rule_dtype.itemsize == 40


def default_sched():
    s = np.zeros((), dtype=sched_dtype)
    s["launch_order"], s["publish_factor"], s["backlog"], s["min_nodes"], s["cloud_blocks"], s["no_child_bound"] = 1, 4, 32, 2, 1, 0
    s["struct_size"] = sched_dtype.itemsize
    return s


def default_params():
    p = np.zeros((), dtype=params_dtype)
    p["feas_tol"] = 1e-9
    p["dep_tol"] = 1e-10
    p["max_nodes"] = 100000
    p["max_iters"] = 2000
    p["max_work"] = 0
    p["share"] = 1
    p["mip_gap"] = 0.0
    p["deadline_ms"] = 0.0
    return p


def ptr(a):
    """void* to the first byte of a C-contiguous numpy array."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ctypes.c_void_p)


def make_problems(n):
    pr = np.zeros(n, dtype=problem_dtype)
    return pr


def pack_faces(polys):
    """polys: list of (A[F,3], b[F]) -> (faces array, face_off list)"""
    off = [0]
    rows = []
    for A, b in polys:
        A = np.asarray(A, dtype=np.float64).reshape(-1, 3)
        b = np.asarray(b, dtype=np.float64).reshape(-1)
        assert A.shape[0] == b.shape[0]
        for i in range(A.shape[0]):
            rows.append((A[i], b[i]))
        off.append(off[-1] + A.shape[0])
    faces = np.zeros(len(rows), dtype=face_dtype)
    for i, (a, bb) in enumerate(rows):
        faces[i]["a"] = a
        faces[i]["b"] = bb
    return faces, off


def set_pins(problem, assign):
    """Fix the binaries of one problem record: assign[t] = polytope index or -1 (free)."""
    w = 0
    for t, a in enumerate(assign):
        if a is not None and a >= 0:
            w |= (int(a) + 1) << (4 * t)
    problem["pin"][0] = w & 0xFFFFFFFF
    problem["pin"][1] = (w >> 32) & 0xFFFFFFFF
