"""Corridor front-end: faster_amd/host/corridor_frontend.hpp against the independent numpy restatement
oracle/decomp_oracle.py, plus the properties the reference relies on (parity unpinned: DecompUtil/jps3d cannot be built here and
their own tests assert nothing)."""
import numpy as np
import pytest

from faster_amd import frontend
from oracle import decomp_oracle


@pytest.fixture(scope="module", autouse=True)
def built():
    from faster_amd import build as fb

    fb.build_frontend()


def random_scene(rng, n_pts=400):
    path = np.cumsum(np.vstack([rng.uniform(-3, 3, 3) * [1, 1, 0] + [0, 0, 1.2], rng.uniform(0.8, 2.5, (3, 1)) * (rng.normal(size=(3, 3)) * [1, 1, 0.2])]), axis=0)
    path[:, 2] = np.clip(path[:, 2], 0.6, 2.4)   # above the ground plane
    cloud = rng.uniform(path.min(0) - 2.5, path.max(0) + 2.5, size=(n_pts, 3))
    # keep the path itself clear by 0.45 m so that every segment has a non-degenerate ellipsoid
    keep = np.ones(len(cloud), bool)
    for a, b in zip(path[:-1], path[1:]):
        t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
        keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > 0.45
    return path, cloud[keep]


@pytest.mark.parametrize("seed", range(6))
def test_decomposition_matches_numpy_restatement(seed):
    rng = np.random.default_rng(seed)
    path, cloud = random_scene(rng)
    got, ell = frontend.decompose(path, cloud, drone_radius=0.05, z_ground=0.0)
    ref = decomp_oracle.decompose_path(path, cloud, 0.05, 0.0)
    assert len(got) == len(ref) == len(path) - 1
    for i, ((A, b), (Ar, br)) in enumerate(zip(got, ref)):
        assert A.shape == Ar.shape, (i, A.shape, Ar.shape)
        # the fitted ellipsoid touches two obstacle points (distance exactly 1 for both), so which of them yields the first
        # separating plane is decided by the last bit: compare the polytopes as sets of rows
        key = lambda M: M[np.lexsort(np.round(M, 7).T[::-1])]
        np.testing.assert_allclose(key(np.column_stack([A, b])), key(np.column_stack([Ar, br])), atol=1e-9)
        _, (Rf, axes, c) = decomp_oracle.decompose_segment(path[i], path[i + 1], cloud, inflate=0.05)
        np.testing.assert_allclose(ell[i, 9:12], axes, atol=1e-9)
        np.testing.assert_allclose(ell[i, 12:15], c, atol=1e-12)


@pytest.mark.parametrize("seed", range(4))
def test_decomposition_properties(seed):
    """Each polytope contains its path segment; no (inflated) obstacle point of its local box lies strictly inside; normals are
    unit; the last row is the ground plane (jps_manager.cpp:113-124); 6 local-box faces are present (line_segment.h:57-98)."""
    rng = np.random.default_rng(100 + seed)
    path, cloud = random_scene(rng, 600)
    infl = 0.05
    got, ell = frontend.decompose(path, cloud, drone_radius=infl, z_ground=0.0)
    for i, (A, b) in enumerate(got):
        np.testing.assert_allclose(np.linalg.norm(A, axis=1), 1.0, atol=1e-12)
        assert np.array_equal(A[-1], [0, 0, -1]) and b[-1] == 0
        for t in np.linspace(0, 1, 11):
            assert np.all(A @ (path[i] + t * (path[i + 1] - path[i])) <= b + 1e-9)
        assert A.shape[0] >= 7
        # obstacle points (as the decomposition inflates them) are never strictly inside
        R, axes, c = ell[i, :9].reshape(3, 3), ell[i, 9:12], ell[i, 12:15]
        Ri = decomp_oracle.rot_x_to(path[i + 1] - path[i])
        loc = (cloud - c) @ Ri
        infl_pts = (loc - np.sign(loc) * infl) @ Ri.T + c
        inside_box = np.all(infl_pts @ A[-7:-1].T <= b[-7:-1] + 1e-9, axis=1) & np.all(cloud @ A[-7:-1].T <= b[-7:-1] + 1e-9, axis=1)
        strictly_inside = np.all(infl_pts[inside_box] @ A[:-1].T < b[:-1] - 1e-7, axis=1)
        assert not strictly_inside.any()
        # and the ellipsoid holds no inflated obstacle point strictly inside
        d = np.linalg.norm(((infl_pts[inside_box] - c) @ R) / axes, axis=1)
        assert np.all(d >= 1 - 1e-7)


def test_empty_cloud_gives_the_local_box():
    path = np.array([[0, 0, 1.0], [3, 0, 1.0]])
    got, _ = frontend.decompose(path, np.zeros((0, 3)))
    A, b = got[0]
    assert A.shape == (7, 3)          # 6 box faces + ground
    assert np.all(A @ np.array([1.5, 0, 1.0]) < b)
    assert np.all(A @ np.array([4.9, 1.9, 1.9]) <= b + 1e-12) and np.any(A @ np.array([5.1, 0, 1.0]) > b)


@pytest.mark.parametrize("seed,start,goal", [(7, (0.8, 0.9, 1.0), (9.1, 9.2, 1.1)), (8, (9.0, 0.8, 0.4), (1.0, 9.3, 1.7)),
                                             (9, (5.0, 0.7, 1.6), (5.2, 9.4, 0.3))])
def test_path_search_properties(seed, start, goal):
    """Voxel path search (exact empty-grid heuristic, total order on the open list): endpoints forced onto start/goal (jps_manager.cpp:175-186), every leg passes jps3d's ray test, and the
    cleaned path is no longer than an independently computed optimal 26-connected grid path (scipy Dijkstra)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra

    cloud, centres = frontend.forest_cloud(seed, size=(10.0, 10.0, 2.0), density=0.15)
    res, infl = 0.25, 0.25
    cells = (44, 44, 8)
    center = np.array([5.0, 5.0, 1.0])
    start, goal = np.array(start), np.array(goal)
    path = frontend.plan(cloud, cells, res, center, 0.0, 2.0, infl, start, goal)
    assert path is not None and len(path) >= 2
    np.testing.assert_allclose(path[0], start)
    np.testing.assert_allclose(path[-1], goal)
    # rebuild the occupancy grid exactly as the front-end does (readMap semantics) and check the legs
    nx, ny = cells[0] + int(5 * infl / res), cells[1] + int(5 * infl / res)
    nz = int(min(cells[2] / 2.0, (center[2] - 0.0) / res)) + max(int((2.0 - center[2]) / res), 1) if True else 0
    origin = np.array([center[0] - res * nx / 2, center[1] - res * ny / 2, center[2] - res * int(min(cells[2] / 2.0, (center[2]) / res))])
    occ = np.zeros((nx, ny, nz), bool)
    m = int(np.floor(infl / res))
    idx = np.maximum(np.round((cloud - origin) / res - 0.5).astype(int), 0)
    for dx in range(-m, m + 1):
        for dy in range(-m, m + 1):
            for dz in range(-m, m + 1):
                j = idx + [dx, dy, dz]
                ok = np.all((j >= 0) & (j < [nx, ny, nz]), axis=1)
                occ[j[ok, 0], j[ok, 1], j[ok, 2]] = True

    def cell(p):
        return np.round((p - origin) / res - 0.5).astype(int)

    mf = int(round(float(np.float32(infl)) / res + 0.5))   # setFreeVoxelAndSurroundings: round(d / res + 0.5), d a float (map_util.h:248-263)
    for c in (cell(start), cell(goal)):
        occ[max(c[0] - mf, 0):c[0] + mf + 1, max(c[1] - mf, 0):c[1] + mf + 1, max(c[2] - mf, 0):c[2] + mf + 1] = False
    for v in path:
        assert not occ[tuple(cell(v))], "path vertex in an occupied cell"
    def ray_clear(a, b):
        steps = int(np.max(np.abs(b - a)) / res / 0.8)  # jps3d's ray test samples every 0.8 cell (map_util.h:348-382)
        for k in range(1, steps):
            c = cell(a + (b - a) * (k / steps))
            if not (np.all(c >= 0) and np.all(c < [nx, ny, nz])):
                break
            if occ[tuple(c)]:
                return False
        return True

    # interior legs (the two end legs are re-anchored on start/goal afterwards).  removeCornerPts runs forwards and then on the
    # reversed path (jps_planner.cpp:286-291) and the samples of a ray depend on the end it starts from: a leg was accepted in one
    # of the two directions
    def grid_run(a, b):   # a straight run of the raw cell path (removeLinePts keeps its ends): unit steps through free cells
        ca, cb = cell(a), cell(b)
        d = cb - ca
        n = int(np.max(np.abs(d)))
        if n == 0 or np.any((np.abs(d) != 0) & (np.abs(d) != n)):
            return False
        return all(not occ[tuple(ca + (d // n) * k)] for k in range(n + 1))

    for a, b in zip(path[1:-2], path[2:-1]):
        assert ray_clear(a, b) or ray_clear(b, a) or grid_run(a, b), "path leg is neither a run of free cells nor an accepted shortcut"
    # optimal grid distance
    free = ~occ
    ids = -np.ones(occ.shape, int)
    ids[free] = np.arange(free.sum())
    rows, cols, w = [], [], []
    fx, fy, fz = np.nonzero(free)
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                if dx == dy == dz == 0:
                    continue
                x, y, z = fx + dx, fy + dy, fz + dz
                ok = (x >= 0) & (x < nx) & (y >= 0) & (y < ny) & (z >= 0) & (z < nz)
                ok[ok] &= free[x[ok], y[ok], z[ok]]
                rows.append(ids[fx[ok], fy[ok], fz[ok]]); cols.append(ids[x[ok], y[ok], z[ok]])
                w.append(np.full(ok.sum(), np.sqrt(dx * dx + dy * dy + dz * dz)))
    g = coo_matrix((np.concatenate(w), (np.concatenate(rows), np.concatenate(cols))), shape=(free.sum(), free.sum())).tocsr()
    dist = dijkstra(g, indices=ids[tuple(cell(start))])[ids[tuple(cell(goal))]]
    length = np.sum(np.linalg.norm(np.diff(path, axis=0), axis=1))
    assert np.isfinite(dist)
    assert length <= dist * res + 2 * res * np.sqrt(3) + 1e-9     # shortcuts only shorten; ends moved off the cell centres
    assert length >= np.linalg.norm(goal - start) - 1e-9


def test_forest_batch_feeds_the_solver(oracle):
    """Config-5 style problems from the front-end are valid solver inputs: x0 inside the first polytope, goal inside the last,
    consecutive polytopes overlap on the shared vertex, and the oracle solves most of them."""
    pr, faces, info = frontend.forest_batch(48, seed=3, n_seg=8, max_poly=5)
    assert len(pr) >= 40 and info["overflow"] == 0
    assert 7 <= info["faces_per_polytope"] <= 40
    for p in pr:
        fb = int(p["face_begin"])
        f0, f1 = fb + p["face_off"][0], fb + p["face_off"][1]
        assert np.all(faces["a"][f0:f1] @ p["x0"][:3] <= faces["b"][f0:f1] + 1e-9)
        k = int(p["n_poly"]) - 1
        f0, f1 = fb + p["face_off"][k], fb + p["face_off"][k + 1]
        assert np.all(faces["a"][f0:f1] @ p["xf"][:3] <= faces["b"][f0:f1] + 1e-9)
    res = oracle.solve_batch(pr, faces)
    assert res["solved"].mean() > 0.7


@pytest.mark.parametrize("seed", [3, 4])
def test_path_search_equals_pure_python_restatement(seed):
    """The C++ front-end's map and path search against an independent pure-Python implementation (oracle/path_oracle.py: lists,
    heapq): same occupancy grid, same vertices, same number of expanded cells — the order on the open list is total, so the container
    cannot matter, and all arithmetic is +, *, /, sqrt on doubles."""
    from oracle import path_oracle

    cloud, centres = frontend.forest_cloud(seed, size=(8.0, 8.0, 2.0), density=0.2)
    res, infl, zg, zmax = 0.25, 0.25, 0.0, 2.0
    cells, center = (34, 34, 8), np.array([4.0, 4.0, 1.0])
    rng = np.random.default_rng(seed)
    n = 10
    starts = np.column_stack([rng.uniform(0.5, 2.5, n), rng.uniform(0.5, 7.5, n), rng.uniform(0.3, 1.7, n)])
    goals = np.column_stack([rng.uniform(5.5, 7.5, n), rng.uniform(0.5, 7.5, n), rng.uniform(0.3, 1.7, n)])
    goals[0] = starts[0] + 0.01            # same cell
    starts[1] = [-20.0, 4.0, 1.0]          # outside the map
    starts[2, 2] = -0.4                    # clamped to the ground
    hp, hn, hex_, occ, dims, origin = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, want_grid=True)
    g = path_oracle.Grid(cloud.tolist(), cells, res, center.tolist(), zg, zmax, infl)
    assert (g.nx, g.ny, g.nz) == tuple(int(v) for v in dims) and np.array_equal(np.array(g.origin), origin)
    assert np.array_equal(np.frombuffer(bytes(g.occ), dtype=np.int8).reshape(occ.shape), occ)
    solved = 0
    for i in range(n):
        path, ex = path_oracle.plan(g, starts[i].tolist(), goals[i].tolist())
        if path is None:
            assert hn[i] == 0, i
            continue
        solved += 1
        assert hn[i] == len(path) and ex == hex_[i], (i, hn[i], len(path), ex, hex_[i])
        assert np.array_equal(np.array(path), hp[i, :hn[i]]), i
    assert solved >= 7


def test_host_settings_come_back():
    """forest_batch(front="host") used to leave the process-wide search mode / sphere radius of the host front-end at
    astar / 0 whatever the caller had set.  The caller's settings survive a call that needs other ones, and an exception inside it."""
    from faster_amd import frontend

    frontend.set_search("jps")
    frontend.set_sphere(3.5)
    try:
        frontend.forest_batch(8, seed=2, n_seg=6, max_poly=3, search="astar", sphere_ra=0.0)
        assert frontend._HOST == {"search": "jps", "sphere": 3.5}
        try:
            with frontend.host_settings("astar", 1.0):
                assert frontend._HOST == {"search": "astar", "sphere": 1.0}
                raise KeyError("inside")
        except KeyError:
            pass
        assert frontend._HOST == {"search": "jps", "sphere": 3.5}
    finally:
        frontend.set_search("astar")
        frontend.set_sphere(0.0)
