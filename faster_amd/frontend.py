"""ctypes binding of the CPU corridor front-end (faster_amd/libfasterfront.so) and the Monte-Carlo
forest workload of BASELINE config 5."""
import contextlib
import ctypes
import os
import threading

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libfasterfront.so")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError("%s is missing: run `python -m faster_amd.build`" % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        vp, i32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.ff_decompose.restype = i32
        L.ff_decompose.argtypes = [vp, i32, vp, i32, vp, f64, f64, vp, i32, vp, vp]
        L.ff_plan.restype = i32
        L.ff_plan.argtypes = [vp, i32, i32, i32, i32, f64, vp, f64, f64, f64, vp, vp, vp, i32]
        L.ff_set_search_mode.restype = i32
        L.ff_set_search_mode.argtypes = [i32]
        L.ff_set_sphere.restype = i32
        L.ff_set_sphere.argtypes = [f64]
        L.ff_plan_jps.restype = i32
        L.ff_plan_jps.argtypes = [vp, i32, i32, i32, i32, f64, vp, f64, f64, f64, vp, vp, vp, i32, vp, vp]
        L.ff_jps_tables.restype = None
        L.ff_jps_tables.argtypes = [vp, vp, vp]
        L.ff_plan_batch.restype = i32
        L.ff_plan_batch.argtypes = [vp, i32, i32, i32, i32, f64, vp, f64, f64, f64, vp, vp, i32, i32, f64, i32, vp, vp, vp, vp, vp, vp]
        L.ff_corridor_batch.restype = i32
        L.ff_corridor_batch.argtypes = [vp, i32, i32, i32, i32, f64, vp, f64, f64, f64, f64, vp, vp, i32, i32, f64, i32, vp, vp, vp, vp]
        _LIB = L
    return _LIB


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def decompose(path, cloud, drone_radius=0.05, z_ground=0.0, bbox=(2.0, 2.0, 1.0), max_faces=4096):
    """JPS_Manager::cvxEllipsoidDecomp for one path. Returns ([(A, b) per segment], ellipsoids[n_seg, 15])."""
    path, cloud, bbox = _c(path).reshape(-1, 3), _c(cloud).reshape(-1, 3), _c(bbox)
    nseg = len(path) - 1
    faces = np.zeros((max_faces, 4))
    off = np.zeros(nseg + 1, dtype=np.int32)
    ell = np.zeros((nseg, 15))
    tot = lib().ff_decompose(abi.ptr(path), len(path), abi.ptr(cloud) if len(cloud) else None, len(cloud), abi.ptr(bbox), drone_radius,
                             z_ground, abi.ptr(faces), max_faces, abi.ptr(off), abi.ptr(ell))
    if tot < 0:
        raise RuntimeError("max_faces too small")
    return [(faces[off[i]:off[i + 1], :3].copy(), faces[off[i]:off[i + 1], 3].copy()) for i in range(nseg)], ell


def plan(cloud, cells, res, center, z_ground, z_max, inflation, start, goal, max_points=4096):
    """JPS_Manager::solveJPS3D (A* variant). Returns the cleaned path [k, 3] or None."""
    cloud = _c(cloud).reshape(-1, 3)
    out = np.zeros((max_points, 3))
    k = lib().ff_plan(abi.ptr(cloud) if len(cloud) else None, len(cloud), int(cells[0]), int(cells[1]), int(cells[2]), res, abi.ptr(_c(center)),
                      z_ground, z_max, inflation, abi.ptr(_c(start)), abi.ptr(_c(goal)), abi.ptr(out), max_points)
    if k < 0:
        raise RuntimeError("max_points too small")
    return out[:k].copy() if k > 0 else None


def plan_jps(cloud, cells, res, center, z_ground, z_max, inflation, start, goal, max_points=4096):
    """JPS_Manager::solveJPS3D with jps3d's own jump point search order (plan_path_jps). -> (path [k, 3] or None, raw cost [m], expansions)"""
    import ctypes

    cloud = _c(cloud).reshape(-1, 3)
    out = np.zeros((max_points, 3))
    cost, ex = ctypes.c_double(0.0), ctypes.c_longlong(0)
    k = lib().ff_plan_jps(abi.ptr(cloud) if len(cloud) else None, len(cloud), int(cells[0]), int(cells[1]), int(cells[2]), res, abi.ptr(_c(center)),
                          z_ground, z_max, inflation, abi.ptr(_c(start)), abi.ptr(_c(goal)), abi.ptr(out), max_points, ctypes.byref(cost), ctypes.byref(ex))
    if k < 0:
        raise RuntimeError("max_points too small")
    return (out[:k].copy() if k > 0 else None), cost.value, ex.value


# The host front-end keeps its search mode and sphere radius as process-wide settings (ff_set_search_mode / ff_set_sphere).  This module
# mirrors them so that a call which needs other values can put the caller's back (host_settings), under a lock: the batch entry
# points read the settings while they run.
_HOST = {"search": "astar", "sphere": 0.0}
_HOST_LOCK = threading.RLock()


def set_search(mode):
    """Which search plan_batch / forest_batch(front="host") run: "astar" (default: the total-order A* the device search reproduces bit for
    bit) or "jps" (jump point search in jps3d's own order: the path FASTER itself gets)."""
    with _HOST_LOCK:
        if lib().ff_set_search_mode({"astar": 0, "jps": 1}[mode]) != 0:
            raise ValueError(mode)
        _HOST["search"] = mode


def set_sphere(ra):
    """Clip every path of plan_batch / forest_batch(front="host") to JPS_in (Faster::replan, faster.cpp:370-382) before the vertex
    refinement: sphere of radius min(|goal - start| - 0.001, ra) around the start; 0: off."""
    with _HOST_LOCK:
        lib().ff_set_sphere(float(ra))
        _HOST["sphere"] = max(float(ra), 0.0)


@contextlib.contextmanager
def host_settings(search, sphere_ra):
    """The host front-end with this search and sphere radius for the duration of the block; the caller's settings come back afterwards,
    also when the block raises.  Holds the lock: concurrent callers with other settings wait."""
    with _HOST_LOCK:
        prev = dict(_HOST)
        try:
            set_search(search)
            set_sphere(sphere_ra)
            yield
        finally:
            set_search(prev["search"])
            set_sphere(prev["sphere"])


def jps_tables():
    """The neighbour tables plan_path_jps generates, in the layout of jps3d's JPS3DNeib."""
    ns, f1, f2 = np.zeros((27, 3, 26), dtype=np.int32), np.zeros((27, 3, 12), dtype=np.int32), np.zeros((27, 3, 12), dtype=np.int32)
    lib().ff_jps_tables(abi.ptr(ns), abi.ptr(f1), abi.ptr(f2))
    return ns, f1, f2


def plan_batch(cloud, cells, res, center, z_ground, z_max, inflation, starts, goals, max_points=64, max_vertex_dist=0.0, max_poly=0,
               want_grid=False):
    """n queries over one map on the CPU (OpenMP): the counterpart of capi.Map.plan_batch, same outputs
    (paths [n][max_points][3], n_points [n], expansions [n]) (+ occupancy [nz][ny][nx], dims, origin with want_grid)."""
    cloud = _c(cloud).reshape(-1, 3)
    starts, goals = _c(starts).reshape(-1, 3), _c(goals).reshape(-1, 3)
    n = len(starts)
    paths = np.zeros((n, max_points, 3))
    npts = np.zeros(n, dtype=np.int32)
    ex = np.zeros(n, dtype=np.int64)
    dims = np.zeros(3, dtype=np.int32)
    origin = np.zeros(3)
    occ = None
    if want_grid:  # geometry first (the grid size is an output)
        lib().ff_plan_batch(abi.ptr(cloud) if len(cloud) else None, 0, int(cells[0]), int(cells[1]), int(cells[2]), res, abi.ptr(_c(center)),
                            z_ground, z_max, inflation, None, None, 0, max_points, 0.0, 0, None, abi.ptr(npts), None, None, abi.ptr(dims),
                            abi.ptr(origin))
        occ = np.zeros(int(dims[0]) * int(dims[1]) * int(dims[2]), dtype=np.int8)
    lib().ff_plan_batch(abi.ptr(cloud) if len(cloud) else None, len(cloud), int(cells[0]), int(cells[1]), int(cells[2]), res,
                        abi.ptr(_c(center)), z_ground, z_max, inflation, abi.ptr(starts), abi.ptr(goals), n, max_points, max_vertex_dist,
                        max_poly, abi.ptr(paths), abi.ptr(npts), abi.ptr(ex), abi.ptr(occ) if occ is not None else None, abi.ptr(dims),
                        abi.ptr(origin))
    if want_grid:
        return paths, npts, ex, occ.reshape(int(dims[2]), int(dims[1]), int(dims[0])), dims, origin
    return paths, npts, ex


def forest_queries(n, seed, size=(20.0, 20.0, 3.0), res=0.2, inflation=0.3, min_goal_dist=6.0, return_rng=False):
    """The map and the start/goal pairs of BASELINE config 5 (what forest_batch searches): cloud, cells, center, starts, goals."""
    rng = np.random.default_rng(seed + 1)
    cloud, centres = forest_cloud(seed, size)

    def free_points(k):
        out = np.zeros((0, 3))
        while len(out) < k:
            p = np.column_stack([rng.uniform(1, size[0] - 1, 2 * k), rng.uniform(1, size[1] - 1, 2 * k), rng.uniform(0.8, size[2] - 0.8, 2 * k)])
            d = np.min(np.linalg.norm(p[:, None, :2] - centres[None, :, :], axis=2), axis=1)
            out = np.vstack([out, p[d > 0.3 + inflation + 0.35]])
        return out[:k]

    starts = free_points(n)
    goals = free_points(n)
    for _ in range(8):  # resample goals that are too close
        near = np.linalg.norm(goals - starts, axis=1) < min_goal_dist
        if not near.any():
            break
        goals[near] = free_points(int(near.sum()))
    cells = (int(size[0] / res) + 10, int(size[1] / res) + 10, int(size[2] / res))
    center = np.array([size[0] / 2, size[1] / 2, size[2] / 2])
    if return_rng:
        return cloud, cells, center, starts, goals, rng
    return cloud, cells, center, starts, goals


def forest_cloud(seed, size=(20.0, 20.0, 3.0), density=0.1, radius=0.3, spacing=0.15):
    """Random forest of vertical cylinders (BASELINE config 5: 20 x 20 x 3 m, r = 0.3 m, 0.1 trees/m^2) sampled as the
    occupied-point cloud a mapper would deliver (surface points every `spacing` m)."""
    rng = np.random.default_rng(seed)
    n_trees = int(round(density * size[0] * size[1]))
    centres = rng.uniform([0.5, 0.5], [size[0] - 0.5, size[1] - 0.5], size=(n_trees, 2))
    ang = np.arange(0, 2 * np.pi, spacing / radius)
    zs = np.arange(0.0, size[2] + 1e-9, spacing)
    ring = np.stack([radius * np.cos(ang), radius * np.sin(ang)], axis=1)
    pts = (centres[:, None, None, :] + ring[None, None, :, :]) + np.zeros((1, len(zs), 1, 1))
    cloud = np.concatenate([pts.reshape(-1, 2), np.tile(np.repeat(zs, len(ang)), n_trees)[:, None]], axis=1)
    return cloud, centres


def corridor_batch_device(ctx, vmap, cloud, cells, res, center, z_max, inflation, starts, goals, max_poly, max_vertex_dist, faces_per_problem,
                          drone_radius, z_ground=0.0, device=0, search="astar", sphere_ra=0.0):
    """The corridor front-end on the device: fh_map_read + fh_map_plan_batch_device (path search, createMoreVertexes,
    deleteVertexes) + fh_corridor_batch_device (decomposition, rows in fh_problem's layout).  `ctx`: capi.Context, `vmap`: capi.Map.
    Returns host copies (faces [n][fpp][4], face_off [n][9], n_poly [n], goal [n][3]) and the device times in seconds."""
    import time

    import torch

    dev = torch.device("cuda", device)
    n, mp = len(starts), max_poly + 1
    d_cloud = torch.from_numpy(_c(cloud).reshape(-1, 3)).to(dev)
    d_s, d_g = torch.from_numpy(_c(starts)).to(dev), torch.from_numpy(_c(goals)).to(dev)
    d_paths = torch.zeros((n, mp, 3), dtype=torch.float64, device=dev)
    d_np = torch.zeros(n, dtype=torch.int32, device=dev)
    d_ex = torch.zeros(n, dtype=torch.int64, device=dev)
    d_faces = torch.zeros((n, faces_per_problem, 4), dtype=torch.float64, device=dev)
    d_off = torch.zeros((n, 9), dtype=torch.int32, device=dev)
    d_npoly = torch.zeros(n, dtype=torch.int32, device=dev)
    d_goal = torch.zeros((n, 3), dtype=torch.float64, device=dev)
    prev = (vmap.search, vmap.sphere)  # the caller's Map comes back as it was (changing the search mode invalidates its workspace:
    try:                               # only touched when it differs)
        if vmap.search != search:
            vmap.set_search(search)
        vmap.set_sphere(sphere_ra)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        vmap.read_device(d_cloud.data_ptr(), len(d_cloud), cells, res, center, z_ground, z_max, inflation)
        vmap.sync()
        t1 = time.perf_counter()
        vmap.plan_batch_device(d_s.data_ptr(), d_g.data_ptr(), n, mp, d_paths.data_ptr(), d_np.data_ptr(), d_ex.data_ptr(), max_vertex_dist, max_poly)
        vmap.sync()
        t2 = time.perf_counter()
    finally:
        if vmap.search != prev[0]:
            vmap.set_search(prev[0])
        vmap.set_sphere(prev[1])
    ctx.corridor_batch_device(d_cloud.data_ptr(), len(d_cloud), d_paths.data_ptr(), d_np.data_ptr(), n, mp, max_poly, faces_per_problem,
                              d_faces.data_ptr(), d_off.data_ptr(), d_npoly.data_ptr(), d_goal.data_ptr(), drone_radius, z_ground)
    ctx.sync()
    t3 = time.perf_counter()
    n_points = d_np.cpu().numpy()   # what the asynchronous planner reported per query: -2 = a limit of the device search
    corridor_batch_device.last_n_points = n_points
    timing = {"map_s": t1 - t0, "path_search_s": t2 - t1, "decomposition_s": t3 - t2, "expansions": int(d_ex.sum().item()),
              "queries_at_a_limit": int((n_points <= -2).sum())}
    goal = d_goal.cpu().numpy()
    goal[np.isnan(goal)] = 0.0
    return d_faces.cpu().numpy(), d_off.cpu().numpy(), d_npoly.cpu().numpy(), goal, timing


def forest_batch(n, seed, n_seg=15, max_poly=8, force_final=True, size=(20.0, 20.0, 3.0), res=0.2, inflation=0.3, drone_radius=0.05,
                 max_vertex_dist=1.5, faces_per_problem=abi.FH_MAX_FACES, min_goal_dist=6.0, front="host", ctx=None, vmap=None, device=0,
                 search="astar", sphere_ra=0.0, **kw):
    """BASELINE config 5: n start/goal pairs in one random forest; corridors from the voxel path search + ellipsoid
    decomposition: front="host" this CPU front-end (OpenMP), front="device" the same steps through the C ABI on the GPU
    (capi.Map + capi.Context; no CPU fallback).  search: "astar" (an optimal path, total order of its own) or "jps" (jump point search in
    jps3d's own order: the path FASTER itself gets); host and device produce the same corridors bit for bit in either.
    Returns (problems, faces, info)."""
    from . import corridor

    cloud, cells, center, starts, goals, rng = forest_queries(n, seed, size, res, inflation, min_goal_dist, return_rng=True)
    if front == "device":
        faces, face_off, n_poly, goal_out, timing = corridor_batch_device(ctx, vmap, cloud, cells, res, center, size[2], inflation, starts, goals,
                                                                          max_poly, max_vertex_dist, faces_per_problem, drone_radius, device=device,
                                                                          search=search, sphere_ra=sphere_ra)
        overflow = 0
    else:
        faces = np.zeros((n, faces_per_problem, 4))
        face_off = np.zeros((n, 9), dtype=np.int32)
        n_poly = np.zeros(n, dtype=np.int32)
        goal_out = np.zeros((n, 3))
        with host_settings(search, sphere_ra):
            overflow = lib().ff_corridor_batch(abi.ptr(_c(cloud)), len(cloud), cells[0], cells[1], cells[2], res, abi.ptr(_c(center)), 0.0, size[2],
                                               inflation, drone_radius, abi.ptr(_c(starts)), abi.ptr(_c(goals)), n, max_poly, max_vertex_dist,
                                               faces_per_problem, abi.ptr(faces), abi.ptr(face_off), abi.ptr(n_poly), abi.ptr(goal_out))
        timing = None
    ok = n_poly > 0
    counts = face_off[np.arange(n), n_poly]
    flat = np.concatenate([faces[i, :counts[i]] for i in np.nonzero(ok)[0]]) if ok.any() else np.zeros((0, 4))
    fc = np.zeros(len(flat), dtype=abi.face_dtype)
    fc["a"], fc["b"] = flat[:, :3], flat[:, 3]
    idx = np.nonzero(ok)[0]
    pr = abi.make_problems(len(idx))
    pr["n_seg"] = n_seg
    pr["n_poly"] = n_poly[idx]
    pr["force_final_pos"] = 1 if force_final else 0
    pr["face_off"] = face_off[idx]
    pr["face_begin"] = np.concatenate([[0], np.cumsum(counts[idx])[:-1]]).astype(np.int32)
    pr["dc"] = 0.01
    pr["v_max"], pr["a_max"], pr["j_max"] = kw.get("v_max", 5.0), kw.get("a_max", 5.0), kw.get("j_max", 8.0)
    pr["f_init"], pr["f_final"], pr["f_inc"] = 1.0, 10.0, 1.0
    u = goal_out[idx] - starts[idx]
    u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-9)
    pr["x0"][:, 0:3] = starts[idx]
    pr["x0"][:, 3:6] = u * rng.uniform(0, 1.5, size=(len(idx), 1))
    pr["xf"][:, 0:3] = goal_out[idx]
    info = {"cloud": cloud, "starts": starts[idx], "goals": goals[idx], "no_path": int((~ok).sum()), "overflow": int(overflow),
            "faces_per_polytope": float(counts[idx].sum() / max(n_poly[idx].sum(), 1)), "front": front, "front_timing": timing,
            "kept": idx}
    return pr, fc, info
