"""ctypes binding of oracle/_ref/libref_frontend.so: the reference's OWN DecompUtil / jps3d sources compiled untouched behind
test-only shims (oracle/ref_frontend/build.sh).  TEST INFRASTRUCTURE: only tests/ may import this module."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(os.path.dirname(HERE), "_ref", "libref_frontend.so")
_LIB = None


def build():
    """Builds the library when /root/reference is present (exit 77 otherwise: the prebuilt file is used). Returns the path or None."""
    r = subprocess.run(["bash", os.path.join(HERE, "build.sh")], capture_output=True, text=True)
    if r.returncode not in (0, 77):
        raise RuntimeError("oracle/ref_frontend/build.sh failed:\n" + r.stderr[-3000:])
    return SO if os.path.exists(SO) else None


def available():
    return os.path.exists(SO)


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(SO)
        vp, i32, f64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
        L.ref_decompose.restype = i32
        L.ref_decompose.argtypes = [vp, i32, vp, i32, f64, f64, vp, vp, i32]
        L.ref_map_create.restype = vp
        L.ref_map_create.argtypes = [vp, i32, i32, i32, i32, f64, vp, f64, f64, f64]
        L.ref_map_destroy.restype = None
        L.ref_map_destroy.argtypes = [vp]
        L.ref_map_dims.restype = None
        L.ref_map_dims.argtypes = [vp, vp, vp]
        L.ref_map_occupancy.restype = None
        L.ref_map_occupancy.argtypes = [vp, vp]
        L.ref_map_plan.restype = i32
        L.ref_map_plan.argtypes = [vp, vp, vp, i32, vp, i32, vp, vp]
        L.ref_frontend_sources.restype = ctypes.c_char_p
        L.ref_jps3d_tables.restype = None
        L.ref_jps3d_tables.argtypes = [vp, vp, vp]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _quiet_stdout:
    """The reference's jps3d prints from inside the library (`ASTAR ERROR!` to stdout for every query without a path,
    thirdparty/jps3d/src/jps_planner/graph_search.cpp:185; planner verbosity): file descriptor 1 points at /dev/null while the
    compiled reference runs, so that its chatter does not bury pytest's summary.  Python-level sys.stdout is untouched."""

    def __enter__(self):
        import sys
        try:
            sys.stdout.flush()
        except Exception:
            pass
        self._saved = os.dup(1)
        self._null = os.open(os.devnull, os.O_WRONLY)
        os.dup2(self._null, 1)

    def __exit__(self, *exc):
        ctypes.CDLL(None).fflush(None)  # the C library's buffered stdout goes to /dev/null too, not to the terminal later
        os.dup2(self._saved, 1)
        os.close(self._saved)
        os.close(self._null)
        return False


def decompose(path, cloud, drone_radius=0.05, z_ground=0.0, max_rows=256):
    """JPS_Manager::cvxEllipsoidDecomp through the reference's EllipsoidDecomp3D: [(A, b)] per leg of the path (ground row last)."""
    path = np.ascontiguousarray(path, dtype=np.float64).reshape(-1, 3)
    cloud = np.ascontiguousarray(cloud, dtype=np.float64).reshape(-1, 3)
    nseg = len(path) - 1
    rows = np.zeros((nseg, max_rows, 4))
    counts = np.zeros(nseg, dtype=np.int32)
    rc = lib().ref_decompose(_p(path), len(path), _p(cloud), len(cloud), drone_radius, z_ground, _p(rows), _p(counts), max_rows)
    if rc != 0:
        raise RuntimeError("max_rows too small")
    return [(rows[i, :counts[i], :3].copy(), rows[i, :counts[i], 3].copy()) for i in range(nseg)]


def jps3d_tables():
    """jps3d's JPS3DNeib tables (ns [27][3][26], f1 [27][3][12], f2 [27][3][12]) as the reference's constructor builds them."""
    ns, f1, f2 = np.zeros((27, 3, 26), dtype=np.int32), np.zeros((27, 3, 12), dtype=np.int32), np.zeros((27, 3, 12), dtype=np.int32)
    lib().ref_jps3d_tables(_p(ns), _p(f1), _p(f2))
    return ns, f1, f2


class Map:
    """JPS_Manager::updateJPSMap (MapUtil::readMap) + solveJPS3D on the reference's jps3d."""

    def __init__(self, cloud, cells, res, center, z_ground, z_max, inflation):
        c = np.ascontiguousarray(cloud, dtype=np.float32).reshape(-1, 3)  # pcl::PointXYZ holds floats
        ce = np.ascontiguousarray(center, dtype=np.float64)
        self._h = lib().ref_map_create(_p(c), len(c), int(cells[0]), int(cells[1]), int(cells[2]), res, _p(ce), z_ground, z_max, inflation)
        self.dims = np.zeros(3, dtype=np.int32)
        self.origin = np.zeros(3)
        lib().ref_map_dims(self._h, _p(self.dims), _p(self.origin))

    def occupancy(self):
        out = np.zeros(int(self.dims[0]) * int(self.dims[1]) * int(self.dims[2]), dtype=np.int8)
        lib().ref_map_occupancy(self._h, _p(out))
        return out.reshape(int(self.dims[2]), int(self.dims[1]), int(self.dims[0]))

    def plan(self, start, goal, use_jps=True, max_pts=4096):
        """(path [k, 3] or None, raw path length in metres, raw path points)"""
        s, g = np.ascontiguousarray(start, dtype=np.float64), np.ascontiguousarray(goal, dtype=np.float64)
        out = np.zeros((max_pts, 3))
        cost = ctypes.c_double(0.0)
        nraw = ctypes.c_int(0)
        with _quiet_stdout():
            k = lib().ref_map_plan(self._h, _p(s), _p(g), 1 if use_jps else 0, _p(out), max_pts, ctypes.byref(cost), ctypes.byref(nraw))
        if k < 0:
            raise RuntimeError("max_pts too small")
        return (out[:k].copy() if k > 0 else None), cost.value, nraw.value

    def close(self):
        if self._h:
            lib().ref_map_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
