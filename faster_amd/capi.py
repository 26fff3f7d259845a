"""ctypes binding of the C ABI (include/fasterhip.h) — the product path.

There is no CPU fallback: if faster_amd/libfasterhip.so is missing this module raises, and every entry
point fails with FH_ERR_DEVICE when no HIP device is present.
"""
import ctypes
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FASTERHIP_SO", os.path.join(_HERE, "libfasterhip.so"))  # override: diagnostic builds only

SYMBOLS = [
    "fh_create", "fh_destroy", "fh_last_error", "fh_default_params", "fh_set_params", "fh_default_sched", "fh_set_sched", "fh_set_stream",
    "fh_request_stop", "fh_clear_stop", "fh_share_stats_read", "fh_share_profile_read", "fh_fp64_peak", "fh_set_pair_margin", "fh_set_pair_rule", "fh_set_unknown_grid_device", "fh_next_goals_device",
    "fh_dt_initial_batch", "fh_dt_initial_batch_device", "fh_solve_batch", "fh_solve_batch_speculative", "fh_solve_batch_device", "fh_sample_batch", "fh_sample_batch_device", "fh_pair_glue_device", "fh_append_plans_device", "fh_safe_corridor_batch_device", "fh_corridor_problems_device", "fh_solve_pairs_device",
    "fh_decompose_batch", "fh_decompose_batch_device", "fh_corridor_batch_device",
    "fh_pool_create", "fh_pool_destroy", "fh_pool_size", "fh_pool_last_error", "fh_pool_set_params", "fh_pool_set_pair_margin", "fh_pool_set_pair_rule", "fh_pool_set_unknown_grid",
    "fh_pool_solve_batch", "fh_pool_solve_pairs",
    "fh_map_create", "fh_map_destroy", "fh_map_last_error", "fh_map_set_stream", "fh_map_set_sched", "fh_map_set_search", "fh_map_set_records", "fh_map_workspace_bytes", "fh_map_set_sphere", "fh_map_sync", "fh_map_read", "fh_map_read_device",
    "fh_map_dims", "fh_map_occupancy", "fh_map_plan_batch", "fh_map_plan_batch_device",
    "fh_sync", "fh_timing_reset", "fh_timing_read", "fh_last_kernel_ms", "fh_last_launch", "fh_version", "fh_abi_version",
    "fh_packed_result_size", "fh_pack_results_device", "fh_pack_results", "fh_unpack_results", "fh_control_points",
]

_LIB = None


class FasterHipError(RuntimeError):
    pass


def packed_result_size(n_seg):
    return int(lib().fh_packed_result_size(int(n_seg)))


def control_points(results, n_seg):
    """fh_control_points: getCP0..3 of every segment of every result -> [n][n_seg][4][3] (zero rows for unsolved results)."""
    results = np.ascontiguousarray(results)
    assert results.dtype == abi.result_dtype
    cp = np.zeros((results.shape[0], int(n_seg), 4, 3), dtype=np.float64)
    rc = lib().fh_control_points(abi.ptr(results), results.shape[0], int(n_seg), abi.ptr(cp))
    if rc != 0:
        raise FasterHipError("fh_control_points: rc=%d" % rc)
    return cp


def pack_results(results, n_seg):
    """numpy array of fh_result records -> uint8 array of packed records (host side)."""
    results = np.ascontiguousarray(results)
    rec = packed_result_size(n_seg)
    if rec <= 0:
        raise FasterHipError("fh_pack_results: n_seg=%r has no packed record (1 <= n_seg <= FH_MAX_SEG)" % (n_seg,))
    if results.dtype != abi.result_dtype:
        raise FasterHipError("fh_pack_results: an array of fh_result records is expected, got dtype %s" % (results.dtype,))
    out = np.zeros(len(results) * rec, dtype=np.uint8)
    rc = lib().fh_pack_results(abi.ptr(results), len(results), int(n_seg), abi.ptr(out))
    if rc != 0:
        raise FasterHipError("fh_pack_results: rc=%d" % rc)
    return out


def unpack_results(packed, n, n_seg):
    """Packed records (bytes-like / uint8 array) -> numpy array of fh_result records (host side, no device needed)."""
    buf = np.ascontiguousarray(np.frombuffer(packed, dtype=np.uint8) if not isinstance(packed, np.ndarray) else packed.view(np.uint8).reshape(-1))
    rec = packed_result_size(n_seg)
    n = int(n)
    if rec <= 0 or n < 0:
        raise FasterHipError("fh_unpack_results: n=%r, n_seg=%r (1 <= n_seg <= FH_MAX_SEG, n >= 0)" % (n, n_seg))
    if buf.size < n * rec:  # a short RCCL / PCIe buffer or a wrong n_seg must not make the C side read past the end
        raise FasterHipError("fh_unpack_results: %d records of %d bytes need %d bytes, the buffer holds %d" % (n, rec, n * rec, buf.size))
    out = np.zeros(n, dtype=abi.result_dtype)
    rc = lib().fh_unpack_results(abi.ptr(buf), n, int(n_seg), abi.ptr(out))
    if rc != 0:
        raise FasterHipError("fh_unpack_results: rc=%d" % rc)
    return out


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(SO_PATH):
            raise FasterHipError("%s is missing: run `python -m faster_amd.build` (hipcc --offload-arch=gfx950). "
                                 "There is no CPU fallback for the hot path." % SO_PATH)
        L = ctypes.CDLL(SO_PATH)
        vp, i32, i64, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
        L.fh_create.restype = i32
        L.fh_create.argtypes = [ctypes.POINTER(vp), i32]
        L.fh_destroy.restype = None
        L.fh_destroy.argtypes = [vp]
        L.fh_last_error.restype = ctypes.c_char_p
        L.fh_last_error.argtypes = [vp]
        L.fh_default_params.restype = None
        L.fh_default_params.argtypes = [vp]
        L.fh_set_params.restype = i32
        L.fh_set_params.argtypes = [vp, vp]
        L.fh_set_stream.restype = i32
        L.fh_set_stream.argtypes = [vp, vp]
        L.fh_default_sched.restype = None
        L.fh_default_sched.argtypes = [vp]
        L.fh_set_sched.restype = i32
        L.fh_set_sched.argtypes = [vp, vp]
        L.fh_map_set_sched.restype = i32
        L.fh_map_set_sched.argtypes = [vp, i32, i32]
        L.fh_map_set_search.restype = i32
        L.fh_map_set_search.argtypes = [vp, i32]
        L.fh_map_set_records.restype = i32
        L.fh_map_set_records.argtypes = [vp, i32]
        L.fh_map_workspace_bytes.restype = ctypes.c_longlong
        L.fh_map_workspace_bytes.argtypes = [vp]
        L.fh_map_set_sphere.restype = i32
        L.fh_map_set_sphere.argtypes = [vp, f64]
        L.fh_set_pair_margin.restype = i32
        L.fh_set_pair_margin.argtypes = [vp, f64]
        L.fh_set_pair_rule.restype = i32
        L.fh_set_pair_rule.argtypes = [vp, vp]
        L.fh_set_unknown_grid_device.restype = i32
        L.fh_set_unknown_grid_device.argtypes = [vp, vp, vp]
        L.fh_next_goals_device.restype = i32
        L.fh_next_goals_device.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
        L.fh_packed_result_size.restype = ctypes.c_size_t
        L.fh_packed_result_size.argtypes = [i32]
        L.fh_pack_results_device.restype = i32
        L.fh_pack_results_device.argtypes = [vp, vp, i32, i32, vp]
        L.fh_unpack_results.restype = i32
        L.fh_unpack_results.argtypes = [vp, i32, i32, vp]
        L.fh_pack_results.restype = i32
        L.fh_pack_results.argtypes = [vp, i32, i32, vp]
        L.fh_request_stop.restype = i32
        L.fh_request_stop.argtypes = [vp]
        L.fh_clear_stop.restype = i32
        L.fh_clear_stop.argtypes = [vp]
        L.fh_share_stats_read.restype = i32
        L.fh_share_stats_read.argtypes = [vp, vp]
        L.fh_share_profile_read.restype = i32
        L.fh_share_profile_read.argtypes = [vp, vp]
        L.fh_fp64_peak.restype = i32
        L.fh_fp64_peak.argtypes = [vp, vp]
        L.fh_solve_batch.restype = i32
        L.fh_solve_batch.argtypes = [vp, vp, vp, i64, i32, vp]
        L.fh_solve_batch_speculative.restype = i32
        L.fh_solve_batch_speculative.argtypes = [vp, vp, vp, i64, i32, i32, vp]
        L.fh_solve_batch_device.restype = i32
        L.fh_solve_batch_device.argtypes = [vp, vp, vp, i32, i32, i32, vp]
        L.fh_sample_batch.restype = i32
        L.fh_sample_batch.argtypes = [vp, vp, vp, i32, i32, vp, vp]
        L.fh_sample_batch_device.restype = i32
        L.fh_sample_batch_device.argtypes = [vp, vp, vp, i32, i32, vp, vp]
        L.fh_pair_glue_device.restype = i32
        L.fh_pair_glue_device.argtypes = [vp, vp, vp, vp, i32, f64, f64, i32, vp, vp]
        L.fh_corridor_problems_device.restype = i32
        L.fh_corridor_problems_device.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
        L.fh_safe_corridor_batch_device.restype = i32
        L.fh_safe_corridor_batch_device.argtypes = [vp, vp, vp, vp, vp, i32, vp, vp, i32, vp, i32, f64, i32, vp, f64, f64, i32, i32, vp, vp, vp, vp]
        L.fh_append_plans_device.restype = i32
        L.fh_append_plans_device.argtypes = [vp, vp, vp, vp, vp, i32, f64, i32, vp, vp, vp]
        L.fh_solve_pairs_device.restype = i32
        L.fh_solve_pairs_device.argtypes = [vp, vp, vp, i32, i32, i32, f64, f64, i32, vp, vp, vp, vp]
        L.fh_decompose_batch.restype = i32
        L.fh_decompose_batch.argtypes = [vp, vp, i32, vp, i32, vp, f64, f64, i32, vp, vp]
        L.fh_decompose_batch_device.restype = i32
        L.fh_decompose_batch_device.argtypes = [vp, vp, i32, vp, i32, vp, f64, f64, i32, vp, vp]
        L.fh_corridor_batch_device.restype = i32
        L.fh_corridor_batch_device.argtypes = [vp, vp, i32, vp, vp, i32, i32, i32, vp, f64, f64, i32, vp, vp, vp, vp]
        L.fh_sync.restype = i32
        L.fh_sync.argtypes = [vp]
        L.fh_pool_create.restype = i32
        L.fh_pool_create.argtypes = [ctypes.POINTER(vp), vp, i32]
        L.fh_pool_destroy.restype = None
        L.fh_pool_destroy.argtypes = [vp]
        L.fh_pool_size.restype = i32
        L.fh_pool_size.argtypes = [vp]
        L.fh_pool_last_error.restype = ctypes.c_char_p
        L.fh_pool_last_error.argtypes = [vp]
        L.fh_pool_set_params.restype = i32
        L.fh_pool_set_params.argtypes = [vp, vp]
        L.fh_pool_set_pair_margin.restype = i32
        L.fh_pool_set_pair_margin.argtypes = [vp, f64]
        L.fh_pool_set_pair_rule.restype = i32
        L.fh_pool_set_pair_rule.argtypes = [vp, vp]
        L.fh_pool_set_unknown_grid.restype = i32
        L.fh_pool_set_unknown_grid.argtypes = [vp, vp, vp]
        L.fh_pool_solve_batch.restype = i32
        L.fh_pool_solve_batch.argtypes = [vp, vp, vp, i64, i32, vp, i32, vp]
        L.fh_pool_solve_pairs.restype = i32
        L.fh_pool_solve_pairs.argtypes = [vp, vp, vp, i64, i32, vp, f64, f64, i32, vp, vp, i32, vp, vp]
        L.fh_map_create.restype = i32
        L.fh_map_create.argtypes = [ctypes.POINTER(vp), i32]
        L.fh_map_destroy.restype = None
        L.fh_map_destroy.argtypes = [vp]
        L.fh_map_last_error.restype = ctypes.c_char_p
        L.fh_map_last_error.argtypes = [vp]
        L.fh_map_set_stream.restype = i32
        L.fh_map_set_stream.argtypes = [vp, vp]
        L.fh_map_sync.restype = i32
        L.fh_map_sync.argtypes = [vp]
        L.fh_map_read.restype = i32
        L.fh_map_read.argtypes = [vp, vp, i32, vp, f64, vp, f64, f64, f64]
        L.fh_map_read_device.restype = i32
        L.fh_map_read_device.argtypes = [vp, vp, i32, vp, f64, vp, f64, f64, f64]
        L.fh_map_dims.restype = i32
        L.fh_map_dims.argtypes = [vp, vp, vp]
        L.fh_map_occupancy.restype = i32
        L.fh_map_occupancy.argtypes = [vp, vp]
        L.fh_map_plan_batch.restype = i32
        L.fh_map_plan_batch.argtypes = [vp, vp, vp, i32, i32, f64, i32, vp, vp, vp]
        L.fh_map_plan_batch_device.restype = i32
        L.fh_map_plan_batch_device.argtypes = [vp, vp, vp, i32, i32, f64, i32, vp, vp, vp]
        L.fh_timing_reset.restype = i32
        L.fh_timing_reset.argtypes = [vp]
        L.fh_timing_read.restype = i32
        L.fh_timing_read.argtypes = [vp, vp, i32]
        L.fh_last_kernel_ms.restype = f64
        L.fh_last_kernel_ms.argtypes = [vp]
        L.fh_dt_initial_batch.restype = i32
        L.fh_dt_initial_batch.argtypes = [vp, vp, i32, vp]
        L.fh_dt_initial_batch_device.restype = i32
        L.fh_dt_initial_batch_device.argtypes = [vp, vp, i32, vp]
        L.fh_last_launch.restype = i32
        L.fh_last_launch.argtypes = [vp, vp]
        L.fh_version.restype = ctypes.c_char_p
        L.fh_abi_version.restype = i32
        L.fh_abi_version.argtypes = []
        if L.fh_abi_version() != abi.FH_ABI_VERSION:
            raise FasterHipError("%s has struct layout generation %d, faster_amd/abi.py describes %d: rebuild (python -m faster_amd.build)"
                                 % (SO_PATH, L.fh_abi_version(), abi.FH_ABI_VERSION))
        _LIB = L
    return _LIB


class Pool:
    """ONE batch sharded over several devices of a node (fh_pool_*): contiguous blocks, host scatter, gather of fh_result blocks."""

    def __init__(self, devices=None):
        self._h = ctypes.c_void_p()
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        rc = lib().fh_pool_create(ctypes.byref(self._h), None if dev is None else abi.ptr(dev), 0 if dev is None else len(dev))
        if rc != 0:
            msg = lib().fh_pool_last_error(self._h).decode() if self._h else "fh_pool_create failed"
            if self._h:
                lib().fh_pool_destroy(self._h)
                self._h = None
            raise FasterHipError("fh_pool_create: rc=%d %s" % (rc, msg))

    def close(self):
        if self._h:
            lib().fh_pool_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return lib().fh_pool_size(self._h)

    def _check(self, rc, what):
        if rc != 0:
            raise FasterHipError("%s: rc=%d %s" % (what, rc, lib().fh_pool_last_error(self._h).decode()))

    def set_params(self, params):
        p = np.ascontiguousarray(params).reshape(1)
        self._check(lib().fh_pool_set_params(self._h, abi.ptr(p)), "fh_pool_set_params")

    def set_pair_margin(self, r_margin):
        self._check(lib().fh_pool_set_pair_margin(self._h, float(r_margin)), "fh_pool_set_pair_margin")

    def set_pair_rule(self, mode=0, r_known=0.0, drone_radius=0.0, delta_h=1.0, delta_a=0.5):
        r = np.zeros(1, dtype=abi.pair_rule_dtype)
        r["mode"], r["r_known"], r["drone_radius"], r["delta_h"], r["delta_a"] = mode, r_known, drone_radius, delta_h, delta_a
        self._check(lib().fh_pool_set_pair_rule(self._h, abi.ptr(r)), "fh_pool_set_pair_rule")

    def set_unknown_grid(self, flags, origin=None, res=None, dims=None):
        """fh_pool_set_unknown_grid: HOST flags [nz][ny][nx] (x fastest), copied to every device of the pool; None: no grid."""
        if flags is None:
            self._check(lib().fh_pool_set_unknown_grid(self._h, None, None), "fh_pool_set_unknown_grid")
            return
        g = np.zeros((), dtype=abi.voxel_grid_dtype)
        g["origin"], g["res"], g["dims"] = origin, res, dims
        g = np.ascontiguousarray(g).reshape(1)
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        self._check(lib().fh_pool_set_unknown_grid(self._h, abi.ptr(g), abi.ptr(flags)), "fh_pool_set_unknown_grid")

    def solve_batch(self, problems, faces, root=0, d_results_root=None):
        problems = np.ascontiguousarray(problems)
        faces = np.ascontiguousarray(faces)
        res = np.zeros(problems.shape[0], dtype=abi.result_dtype)
        self._check(lib().fh_pool_solve_batch(self._h, abi.ptr(problems), abi.ptr(faces) if faces.shape[0] else None, faces.shape[0],
                                              problems.shape[0], abi.ptr(res), root, d_results_root), "fh_pool_solve_batch")
        return res

    def solve_pairs(self, whole, faces, safe_templates, r_frac=0.5, shrink=0.2, max_safe_poly=3, root=0, d_whole_root=None, d_safe_root=None):
        whole = np.ascontiguousarray(whole)
        faces = np.ascontiguousarray(faces)
        safe_templates = np.ascontiguousarray(safe_templates)
        wres = np.zeros(whole.shape[0], dtype=abi.result_dtype)
        sres = np.zeros(whole.shape[0], dtype=abi.result_dtype)
        self._check(lib().fh_pool_solve_pairs(self._h, abi.ptr(whole), abi.ptr(faces) if faces.shape[0] else None, faces.shape[0],
                                              whole.shape[0], abi.ptr(safe_templates), r_frac, shrink, max_safe_poly, abi.ptr(wres),
                                              abi.ptr(sres), root, d_whole_root, d_safe_root), "fh_pool_solve_pairs")
        return wres, sres


class Map:
    """Device voxel map + batched path search (fh_map_*): JPS_Manager::updateJPSMap / solveJPS3D for batches of queries."""

    def __init__(self, device=0):
        self._h = ctypes.c_void_p()
        rc = lib().fh_map_create(ctypes.byref(self._h), device)
        if rc != 0:
            self._h = None
            raise FasterHipError("fh_map_create: rc=%d (no HIP device? there is no CPU fallback)" % rc)
        self.search, self.sphere = "astar", 0.0  # what fh_map_create starts with (mirrored: callers that change them put them back)

    def close(self):
        if self._h:
            lib().fh_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FasterHipError("%s: rc=%d %s" % (what, rc, lib().fh_map_last_error(self._h).decode()))

    def set_stream(self, stream):
        self._check(lib().fh_map_set_stream(self._h, stream), "fh_map_set_stream")

    def sync(self):
        self._check(lib().fh_map_sync(self._h), "fh_map_sync")

    def set_sched(self, waves_per_cu=0, launch_order=1):
        """fh_map_set_sched: wavefronts per CU that hold a search workspace (0 = default 12), far-apart pairs first (default 1)."""
        self._check(lib().fh_map_set_sched(self._h, int(waves_per_cu), int(launch_order)), "fh_map_set_sched")

    def set_sphere(self, ra):
        """fh_map_set_sphere: clip every path to JPS_in (sphere of radius min(|goal - start| - 0.001, ra) around the start); 0: off."""
        self._check(lib().fh_map_set_sphere(self._h, float(ra)), "fh_map_set_sphere")
        self.sphere = max(float(ra), 0.0)

    def set_search(self, mode):
        """fh_map_set_search: "astar" (default; an optimal path, total order of its own) or "jps" (jump point search in jps3d's own
        order: the optimal path FASTER itself gets)."""
        self._check(lib().fh_map_set_search(self._h, {"astar": 0, "jps": 1}[mode]), "fh_map_set_search")
        self.search = mode

    def set_records(self, slots):
        """fh_map_set_records: 0 = one cell record per cell of the map and wavefront; a power of two = that many hashed records per
        wavefront (jump point search only; a query that reaches more than 3/4 of them returns -2); -1 (default) = by the size of the map."""
        self._check(lib().fh_map_set_records(self._h, int(slots)), "fh_map_set_records")

    def workspace_bytes(self):
        """fh_map_workspace_bytes: the search workspace as allocated by the last search."""
        return int(lib().fh_map_workspace_bytes(self._h))

    def read(self, cloud, cells, res, center, z_ground, z_max, inflation):
        cloud = np.ascontiguousarray(cloud, dtype=np.float64).reshape(-1, 3)
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        center = np.ascontiguousarray(center, dtype=np.float64)
        self._check(lib().fh_map_read(self._h, abi.ptr(cloud) if len(cloud) else None, len(cloud), abi.ptr(cells), float(res), abi.ptr(center),
                                      float(z_ground), float(z_max), float(inflation)), "fh_map_read")

    def read_device(self, d_cloud, n_cloud, cells, res, center, z_ground, z_max, inflation):
        cells = np.ascontiguousarray(cells, dtype=np.int32)
        center = np.ascontiguousarray(center, dtype=np.float64)
        self._check(lib().fh_map_read_device(self._h, d_cloud, n_cloud, abi.ptr(cells), float(res), abi.ptr(center), float(z_ground),
                                             float(z_max), float(inflation)), "fh_map_read_device")

    def dims(self):
        d = np.zeros(3, dtype=np.int32)
        o = np.zeros(3, dtype=np.float64)
        self._check(lib().fh_map_dims(self._h, abi.ptr(d), abi.ptr(o)), "fh_map_dims")
        return d, o

    def occupancy(self):
        d, _ = self.dims()
        occ = np.zeros(int(d[0]) * int(d[1]) * int(d[2]), dtype=np.int8)
        self._check(lib().fh_map_occupancy(self._h, abi.ptr(occ)), "fh_map_occupancy")
        return occ.reshape(int(d[2]), int(d[1]), int(d[0]))

    def plan_batch(self, starts, goals, max_points=64, max_vertex_dist=0.0, max_poly=0):
        """-> (paths [n][max_points][3], n_points [n], expansions [n])"""
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        goals = np.ascontiguousarray(goals, dtype=np.float64).reshape(-1, 3)
        n = len(starts)
        paths = np.zeros((n, max_points, 3), dtype=np.float64)
        npts = np.zeros(n, dtype=np.int32)
        ex = np.zeros(n, dtype=np.int64)
        self._check(lib().fh_map_plan_batch(self._h, abi.ptr(starts), abi.ptr(goals), n, max_points, float(max_vertex_dist), int(max_poly),
                                            abi.ptr(paths), abi.ptr(npts), abi.ptr(ex)), "fh_map_plan_batch")
        return paths, npts, ex

    def plan_batch_device(self, d_starts, d_goals, n, max_points, d_paths, d_n_points, d_expansions=None, max_vertex_dist=0.0, max_poly=0):
        self._check(lib().fh_map_plan_batch_device(self._h, d_starts, d_goals, n, max_points, float(max_vertex_dist), int(max_poly), d_paths,
                                                   d_n_points, d_expansions), "fh_map_plan_batch_device")


class Context:
    """One solver context (one HIP stream). Mirrors one SolverGurobi object's GRBEnv/GRBModel."""

    def __init__(self, device=-1, pair_outputs=True, compact_results=False):
        """pair_outputs / compact_results: fh_sched.pair_outputs / .compact_results of this context, kept across set_sched calls.  The
        LIBRARY's defaults are 0 / 0 (a fused pair launch writes the safe problems to memory only when it has to, every word of every
        result is written); this wrapper — the tests and tools read d_safe / d_safe_faces back — asks for complete pair outputs unless
        told otherwise.  bench.py's timed pipelines run pair_outputs=False, compact_results=True: what a streaming caller sets."""
        self._h = ctypes.c_void_p()
        rc = lib().fh_create(ctypes.byref(self._h), device)
        if rc != 0:
            msg = lib().fh_last_error(self._h).decode() if self._h else "fh_create failed"
            if self._h:
                lib().fh_destroy(self._h)
                self._h = None
            raise FasterHipError("fh_create: rc=%d %s" % (rc, msg))
        self._sticky = {"pair_outputs": 1 if pair_outputs else 0, "compact_results": 1 if compact_results else 0}
        if pair_outputs or compact_results:
            self.set_sched()

    def close(self):
        if self._h:
            lib().fh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise FasterHipError("%s: rc=%d %s" % (what, rc, lib().fh_last_error(self._h).decode()))

    def set_params(self, params):
        p = np.ascontiguousarray(params).reshape(1)
        self._check(lib().fh_set_params(self._h, abi.ptr(p)), "fh_set_params")

    def set_stream(self, stream_ptr):
        self._check(lib().fh_set_stream(self._h, ctypes.c_void_p(stream_ptr)), "fh_set_stream")

    def set_sched(self, **kw):
        """fh_set_sched: scheduling of a solve launch (launch_order, publish_factor, backlog, waiting_workgroups, min_nodes,
        cloud_blocks, workgroups_per_cu, child_bound); unnamed fields keep their defaults.  No result field depends on them.
        workgroups_per_cu in 1..8 also selects the kernel build for two wavefronts per SIMD (a batch alone on the device is done sooner)."""
        s = abi.default_sched()
        for k in ("pair_outputs", "compact_results"):
            if k in kw:
                self._sticky[k] = 1 if kw[k] else 0
            s[k] = self._sticky[k]
        for k, v in kw.items():
            if k == "child_bound":   # the field is fh_sched.no_child_bound (0 = default: the bound is on)
                s["no_child_bound"] = 0 if v else 1
            elif k not in self._sticky:
                s[k] = v
        s = np.ascontiguousarray(s).reshape(1)
        self._check(lib().fh_set_sched(self._h, abi.ptr(s)), "fh_set_sched")

    def sync(self):
        self._check(lib().fh_sync(self._h), "fh_sync")

    def set_pair_margin(self, r_margin):
        """r_margin >= 0: the synthetic hand-off keeps R strictly inside its safe corridor (see fasterhip.h); < 0: SURVEY 8(d) literal."""
        self._check(lib().fh_set_pair_margin(self._h, float(r_margin)), "fh_set_pair_margin")

    def set_pair_rule(self, mode=0, r_known=0.0, drone_radius=0.0, delta_h=1.0, delta_a=0.5):
        """fh_set_pair_rule: mode 0 = R at the fraction r_frac of the whole trajectory; 1 = FASTER's findIndexH / findIndexR against
        modelled unknown space (farther than r_known from the start); 2 = the same against the unknown voxels given with
        set_unknown_grid_device (unknown space as an input)."""
        r = np.zeros(1, dtype=abi.pair_rule_dtype)
        r["mode"], r["r_known"], r["drone_radius"], r["delta_h"], r["delta_a"] = mode, r_known, drone_radius, delta_h, delta_a
        self._check(lib().fh_set_pair_rule(self._h, abi.ptr(r)), "fh_set_pair_rule")

    def set_unknown_grid_device(self, d_flags, origin=None, res=None, dims=None):
        """fh_set_unknown_grid_device: the mapper's unknown voxels (rule mode 2): device pointer to dims[0] * dims[1] * dims[2] bytes, x
        fastest, non-zero = unknown; cell centres (i + 0.5) res + origin.  d_flags = None: no unknown grid."""
        if d_flags is None:
            self._check(lib().fh_set_unknown_grid_device(self._h, None, None), "fh_set_unknown_grid_device")
            return
        g = np.zeros((), dtype=abi.voxel_grid_dtype)
        g["origin"], g["res"], g["dims"] = origin, res, dims
        g = np.ascontiguousarray(g).reshape(1)
        self._check(lib().fh_set_unknown_grid_device(self._h, abi.ptr(g), d_flags), "fh_set_unknown_grid_device")

    def pack_results_device(self, d_results, n, n_seg, d_packed):
        """fh_pack_results_device: n fh_result records -> n packed records of packed_result_size(n_seg) bytes (device pointers)."""
        self._check(lib().fh_pack_results_device(self._h, d_results, n, n_seg, d_packed), "fh_pack_results_device")

    def request_stop(self):
        """StopExecution(): callable from any thread while a launch is running."""
        self._check(lib().fh_request_stop(self._h), "fh_request_stop")

    def clear_stop(self):
        self._check(lib().fh_clear_stop(self._h), "fh_clear_stop")

    def share_stats(self):
        """Work-sharing statistics of the most recent solve launch (synchronises)."""
        st = np.zeros((), dtype=abi.share_stats_dtype)
        self._check(lib().fh_share_stats_read(self._h, st.ctypes.data_as(ctypes.c_void_p)), "fh_share_stats_read")
        return {k: int(st[k]) for k in st.dtype.names}

    def share_profile(self):
        out = np.zeros(16, dtype=np.uint64)
        self._check(lib().fh_share_profile_read(self._h, abi.ptr(out)), "fh_share_profile_read")
        return out

    def fp64_peak_tflops(self):
        out = ctypes.c_double(0.0)
        self._check(lib().fh_fp64_peak(self._h, ctypes.byref(out)), "fh_fp64_peak")
        return out.value

    def timing_reset(self):
        self._check(lib().fh_timing_reset(self._h), "fh_timing_reset")

    def timing_read(self, cap=4096):
        """Durations (ms) of the solve-kernel launches since timing_reset(), from HIP events on the stream."""
        ms = np.zeros(cap, dtype=np.float64)
        cnt = lib().fh_timing_read(self._h, abi.ptr(ms), cap)
        if cnt < 0:
            self._check(cnt, "fh_timing_read")
        return ms[: min(cnt, cap)]

    def last_kernel_ms(self):
        return lib().fh_last_kernel_ms(self._h)

    def last_launch(self):
        """fh_last_launch: which solve kernel the most recent solve launch ran — (dict of the fields, the name as rocprofv3 prints it)."""
        info = np.zeros(1, dtype=abi.launch_info_dtype)
        self._check(lib().fh_last_launch(self._h, abi.ptr(info)), "fh_last_launch")
        d = {k: int(info[0][k]) for k in info.dtype.names}
        return d, "fh::solve_kernel<%d, %s, %d, %s>" % (d["n_seg"], "true" if d["pairs"] else "false", d["waves_per_simd"],
                                                      "true" if d["unknown_space"] else "false")

    # ---- host-pointer entry points (numpy in, numpy out) ----
    def dt_initial_batch(self, problems):
        """fh_dt_initial_batch: SolverGurobi::getDTInitial per problem."""
        problems = np.ascontiguousarray(problems)
        assert problems.dtype == abi.problem_dtype
        dt = np.zeros(problems.shape[0], dtype=np.float64)
        self._check(lib().fh_dt_initial_batch(self._h, abi.ptr(problems), problems.shape[0], abi.ptr(dt)), "fh_dt_initial_batch")
        return dt

    def solve_batch(self, problems, faces):
        problems = np.ascontiguousarray(problems)
        faces = np.ascontiguousarray(faces)
        assert problems.dtype == abi.problem_dtype and faces.dtype == abi.face_dtype
        res = np.zeros(problems.shape[0], dtype=abi.result_dtype)
        fptr = abi.ptr(faces) if faces.shape[0] else None
        self._check(lib().fh_solve_batch(self._h, abi.ptr(problems), fptr, faces.shape[0], problems.shape[0], abi.ptr(res)),
                    "fh_solve_batch")
        return res

    def solve_batch_speculative(self, problems, faces, width):
        """fh_solve_batch with the factor line search run `width` factors at a time (same results, lower latency)."""
        problems = np.ascontiguousarray(problems)
        faces = np.ascontiguousarray(faces)
        assert problems.dtype == abi.problem_dtype and faces.dtype == abi.face_dtype
        res = np.zeros(problems.shape[0], dtype=abi.result_dtype)
        fptr = abi.ptr(faces) if faces.shape[0] else None
        self._check(lib().fh_solve_batch_speculative(self._h, abi.ptr(problems), fptr, faces.shape[0], problems.shape[0],
                                                     int(width), abi.ptr(res)), "fh_solve_batch_speculative")
        return res

    def sample_batch(self, problems, results, max_samples):
        problems = np.ascontiguousarray(problems)
        results = np.ascontiguousarray(results)
        n = problems.shape[0]
        states = np.zeros((n, max_samples), dtype=abi.state_dtype)
        counts = np.zeros(n, dtype=np.int32)
        self._check(lib().fh_sample_batch(self._h, abi.ptr(problems), abi.ptr(results), n, max_samples, abi.ptr(states),
                                          abi.ptr(counts)), "fh_sample_batch")
        return states, counts

    def decompose_batch(self, cloud, segments, drone_radius=0.05, z_ground=0.0, bbox=(2.0, 2.0, 1.0), max_faces=64):
        """cvxEllipsoidDecomp on the device for [n, 6] segments sharing one cloud. Returns (faces[n, max_faces], counts[n])."""
        cloud = np.ascontiguousarray(cloud, dtype=np.float64).reshape(-1, 3)
        segments = np.ascontiguousarray(segments, dtype=np.float64).reshape(-1, 6)
        bbox = np.ascontiguousarray(bbox, dtype=np.float64)
        n = segments.shape[0]
        faces = np.zeros((n, max_faces), dtype=abi.face_dtype)
        counts = np.zeros(n, dtype=np.int32)
        self._check(lib().fh_decompose_batch(self._h, abi.ptr(cloud) if len(cloud) else None, len(cloud), abi.ptr(segments), n, abi.ptr(bbox),
                                             drone_radius, z_ground, max_faces, abi.ptr(faces), abi.ptr(counts)), "fh_decompose_batch")
        return faces, counts

    # ---- device-pointer entry points (raw addresses; memory owned by the caller, e.g. torch tensors) ----
    def corridor_batch_device(self, d_cloud, n_cloud, d_paths, d_n_points, n, max_points, max_poly, faces_per_problem, d_faces, d_face_off,
                              d_n_poly, d_goal=None, drone_radius=0.05, z_ground=0.0, bbox=(2.0, 2.0, 1.0)):
        bbox = np.ascontiguousarray(bbox, dtype=np.float64)
        self._check(lib().fh_corridor_batch_device(self._h, d_cloud, n_cloud, d_paths, d_n_points, n, max_points, max_poly, abi.ptr(bbox),
                                                   float(drone_radius), float(z_ground), faces_per_problem, d_faces, d_face_off, d_n_poly,
                                                   d_goal), "fh_corridor_batch_device")

    def solve_batch_device(self, d_problems, d_faces, n, max_seg, max_faces, d_results):
        self._check(lib().fh_solve_batch_device(self._h, d_problems, d_faces, n, max_seg, max_faces, d_results),
                    "fh_solve_batch_device")

    def sample_batch_device(self, d_problems, d_results, n, max_samples, d_states, d_counts):
        self._check(lib().fh_sample_batch_device(self._h, d_problems, d_results, n, max_samples, d_states, d_counts),
                    "fh_sample_batch_device")

    def solve_pairs_device(self, d_whole, d_faces, n, max_seg, max_faces, r_frac, shrink, max_safe_poly, d_whole_results, d_safe,
                           d_safe_faces, d_safe_results):
        self._check(lib().fh_solve_pairs_device(self._h, d_whole, d_faces, n, max_seg, max_faces, r_frac, shrink, max_safe_poly,
                                                d_whole_results, d_safe, d_safe_faces, d_safe_results), "fh_solve_pairs_device")

    def corridor_problems_device(self, d_n_points, d_last_vertex, d_goals, d_faces, d_face_off, d_n_poly, n, faces_per_problem, n_seg, d_problems):
        """fh_corridor_problems_device: polytope table and xf of every problem record from fh_corridor_batch_device's outputs."""
        self._check(lib().fh_corridor_problems_device(self._h, d_n_points, d_last_vertex, d_goals, d_faces, d_face_off, d_n_poly, n, faces_per_problem,
                                                      n_seg, d_problems), "fh_corridor_problems_device")

    def safe_corridor_batch_device(self, d_whole, d_whole_results, d_paths, d_n_points, max_points, d_goals, d_cloud, n_cloud, grid_origin, grid_res,
                                   grid_dims, n, r_frac, max_poly_safe, local_bbox, drone_radius, z_ground, faces_per_problem, n_seg_safe, d_safe,
                                   d_safe_faces, d_safe_paths=None, d_safe_n_points=None):
        """fh_safe_corridor_batch_device: the safe corridor of Faster::replan decomposed around R (unknown space modelled)."""
        g = np.zeros((), dtype=abi.voxel_grid_dtype)
        g["origin"], g["res"], g["dims"] = grid_origin, grid_res, grid_dims
        g = np.ascontiguousarray(g).reshape(1)
        bb = np.ascontiguousarray(local_bbox, dtype=np.float64)
        self._check(lib().fh_safe_corridor_batch_device(self._h, d_whole, d_whole_results, d_paths, d_n_points, max_points, d_goals, d_cloud, n_cloud,
                                                        abi.ptr(g), n, r_frac, max_poly_safe, abi.ptr(bb), drone_radius, z_ground, faces_per_problem,
                                                        n_seg_safe, d_safe, d_safe_faces, d_safe_paths, d_safe_n_points), "fh_safe_corridor_batch_device")

    def append_plans_device(self, d_whole, d_whole_results, d_safe, d_safe_results, n, r_frac, max_states, d_plans, d_counts, d_k_safe=None):
        """fh_append_plans_device: Faster::appendToPlan for a batch of pairs (whole samples 0..k_safe, then the safe samples)."""
        self._check(lib().fh_append_plans_device(self._h, d_whole, d_whole_results, d_safe, d_safe_results, n, r_frac, max_states, d_plans,
                                                 d_counts, d_k_safe), "fh_append_plans_device")

    def next_goals_device(self, d_plans, d_counts, d_cursor, n, max_states, ticks, d_goals, d_ok=None):
        """fh_next_goals_device: Faster::getNextGoal (without yaw) for a batch of committed plans; `ticks` calls in a row."""
        self._check(lib().fh_next_goals_device(self._h, d_plans, d_counts, d_cursor, n, max_states, ticks, d_goals, d_ok), "fh_next_goals_device")

    def pair_glue_device(self, d_whole, d_whole_results, d_faces, n, r_frac, shrink, max_safe_poly, d_safe, d_safe_faces):
        self._check(lib().fh_pair_glue_device(self._h, d_whole, d_whole_results, d_faces, n, r_frac, shrink, max_safe_poly,
                                              d_safe, d_safe_faces), "fh_pair_glue_device")
