"""Row N1 pinned to the reference's OWN code: oracle/_ref/libref_frontend.so is the untouched DecompUtil headers and jps3d
sources of /root/reference/thirdparty compiled behind test-only shims of Eigen / Boost.Heap / ROS / PCL (oracle/ref_frontend/).
The host restatement of the front-end (faster_amd/host/corridor_frontend.cpp, what the device path is checked against bit for
bit) and the numpy restatement (oracle/decomp_oracle.py) are compared with it here; tests/test_gpu_round3.py does the same for
the device kernels.  What is pinned: the occupancy grid (MapUtil::readMap), the polytopes of cvxEllipsoidDecomp row for row, the
COST of the voxel path.  What is measured and reported, not asserted equal: how often the cleaned vertex list equals jps3d's
(FASTER plans with jump point search and a tolerance comparator whose tie order depends on the heap; this repository plans with
A*, an exact heuristic and a total order — equal cost, often another of the equal-cost paths)."""
import json
import os

import numpy as np
import pytest

from faster_amd import build as fb
from faster_amd import frontend

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ref():
    from oracle.ref_frontend import ref as r

    if r.build() is None:
        pytest.skip("oracle/_ref/libref_frontend.so is not built and /root/reference is not present")
    fb.build_frontend()
    return r


def scene(seed, n_legs=4, n_cloud=2500, clearance=0.45):
    rng = np.random.default_rng(seed)
    path = np.cumsum(np.vstack([rng.uniform(-3, 3, 3) * [1, 1, 0] + [0, 0, 1.2], rng.uniform(0.8, 2.5, (n_legs, 1)) * (rng.normal(size=(n_legs, 3)) * [1, 1, 0.2])]), axis=0)
    path[:, 2] = np.clip(path[:, 2], 0.6, 2.4)
    cloud = rng.uniform(path.min(0) - 2.5, path.max(0) + 2.5, size=(n_cloud, 3))
    keep = np.ones(len(cloud), bool)
    for a, b in zip(path[:-1], path[1:]):
        t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
        keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > clearance
    return path, cloud[keep]


def same_rows(got, want, atol=1e-9):
    """Same polytopes as SETS of rows.  (The order of the first planes is not comparable: the fitted ellipsoid touches two or three
    obstacle points at distance exactly 1 by construction, and which of them Ellipsoid::closest_point meets first is decided by the
    last bit of that distance — Eigen's, the shim's and this repository's roundings may differ there.  The set is what constrains.)
    Returns the number of polytopes whose rows also come in the same order."""
    key = lambda M: M[np.lexsort(np.round(M, 6).T[::-1])]
    assert len(got) == len(want)
    ordered = 0
    for i, ((A, b), (A2, b2)) in enumerate(zip(got, want)):
        assert len(b) == len(b2), (i, len(b), len(b2))
        G, W = np.column_stack([A, b]), np.column_stack([A2, b2])
        np.testing.assert_allclose(key(G), key(W), rtol=0, atol=atol, err_msg="polytope %d" % i)
        ordered += int(np.allclose(G, W, rtol=0, atol=atol))
    return ordered


def test_decomposition_equals_the_reference_on_its_own_test_path(ref):
    """The path of decomp_test_node/data/path3d.txt (the provenance of the reference's only corridor fixture) in a random cloud:
    host restatement and numpy restatement give the reference's rows, in the reference's order."""
    from oracle import decomp_oracle

    path = np.array([[5, 11.5, 0.5], [13, 11.5, 3.0], [14, 10.5, 1.5], [14, 5, 2.5]])  # decomp_test_node/data/path3d.txt
    rng = np.random.default_rng(1)
    cloud = rng.uniform(path.min(0) - 3, path.max(0) + 3, size=(4000, 3))
    keep = np.ones(len(cloud), bool)
    for a, b in zip(path[:-1], path[1:]):
        t = np.clip(((cloud - a) @ (b - a)) / ((b - a) @ (b - a)), 0, 1)
        keep &= np.linalg.norm(cloud - (a + t[:, None] * (b - a)), axis=1) > 0.5
    cloud = cloud[keep]
    want = ref.decompose(path, cloud, 0.05, 0.0)
    assert min(len(b) for _, b in want) >= 8
    same_rows(frontend.decompose(path, cloud, 0.05, 0.0)[0], want)
    same_rows(decomp_oracle.decompose_path(path, cloud, 0.05, 0.0), want)


@pytest.mark.parametrize("seed", range(8))
def test_decomposition_equals_the_reference_random_scenes(ref, seed):
    from oracle import decomp_oracle

    path, cloud = scene(100 + seed, n_legs=3 + seed % 4, clearance=0.3 + 0.05 * (seed % 3))
    radius = (0.05, 0.0, 0.2)[seed % 3]
    want = ref.decompose(path, cloud, radius, 0.0)
    same_rows(frontend.decompose(path, cloud, radius, 0.0)[0], want)
    same_rows(decomp_oracle.decompose_path(path, cloud, radius, 0.0), want)


def test_decomposition_equals_the_reference_without_obstacles(ref):
    path = np.array([[0.0, 0.0, 1.0], [2.0, 0.5, 1.2], [2.0, 0.5, 2.2]])   # (the last leg is vertical: dir_h falls back to -x)
    want = ref.decompose(path, np.zeros((0, 3)), 0.05, 0.0)
    same_rows(frontend.decompose(path, np.zeros((0, 3)), 0.05, 0.0)[0], want)
    assert all(len(b) == 7 for _, b in want)   # the local box and the ground


def forest(seed, n, res=0.2, inflation=0.3, size=(20.0, 20.0, 3.0)):
    cloud, cells, center, starts, goals = frontend.forest_queries(n, seed, size=size, res=res, inflation=inflation)
    cloud = cloud.astype(np.float32).astype(np.float64)  # a pcl::PointXYZ cloud holds floats: same points on both sides
    return cloud, cells, center, starts, goals


@pytest.mark.parametrize("seed,res,inflation,z_ground,z_max", [(3, 0.2, 0.3, 0.0, 3.0), (4, 0.25, 0.25, 0.0, 2.0), (5, 0.15, 0.3, 0.4, 2.2)])
def test_occupancy_grid_equals_map_util_read_map(ref, seed, res, inflation, z_ground, z_max):
    """MapUtil::readMap (jps_collision/map_util.h:30-185) itself: same dimensions (the z clipping by z_ground / z_max included), same
    origin, same occupied cells."""
    cloud, cells, center, starts, goals = forest(seed, 8, res=res, inflation=inflation)
    m = ref.Map(cloud, cells, res, center, z_ground, z_max, inflation)
    _, _, _, occ, dims, origin = frontend.plan_batch(cloud, cells, res, center, z_ground, z_max, inflation, starts, goals, max_points=256, want_grid=True)
    assert np.array_equal(m.dims, dims)
    np.testing.assert_allclose(m.origin, origin, rtol=0, atol=1e-12)
    assert np.array_equal(m.occupancy() > 0, occ > 0)
    assert (occ > 0).sum() > 1000
    m.close()


def test_path_cost_equals_jps3d_and_vertex_lists_are_compared(ref):
    """256 forest queries: a path exists for jps3d iff it exists here; the reference's jump point search, the reference's A* and an
    independent Dijkstra agree on the COST of the raw path, and so does this repository's search (same optimum, test_frontend.py);
    the cleaned vertex lists are equal for a part of the queries only — the fraction is written to profiles/ as a measured fact."""
    cloud, cells, center, starts, goals = forest(3, 256)
    res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
    hp, hn, _ = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, max_points=256)
    m = ref.Map(cloud, cells, res, center, zg, zmax, infl)
    same = with_path = 0
    len_ratio = []
    for i in range(len(starts)):
        p, cost, nraw = m.plan(starts[i], goals[i], True)
        pa, cost_a, _ = m.plan(starts[i], goals[i], False)
        assert (p is None) == (hn[i] == 0) == (pa is None), i
        if p is None:
            continue
        with_path += 1
        assert cost == pytest.approx(cost_a, abs=1e-9), i          # jps3d: JPS and A* find paths of one cost
        h = hp[i, :hn[i]]
        np.testing.assert_allclose(h[0], p[0], atol=1e-12)
        np.testing.assert_allclose(h[-1], p[-1], atol=1e-12)
        if len(p) == len(h) and np.allclose(p, h, atol=1e-9):
            same += 1
        len_ratio.append(np.linalg.norm(np.diff(h, axis=0), axis=1).sum() / np.linalg.norm(np.diff(p, axis=0), axis=1).sum())
    m.close()
    assert with_path >= 250
    frac = same / with_path
    out = {"queries": with_path, "identical_vertex_lists": same, "fraction": frac, "cleaned_length_ratio_mean": float(np.mean(len_ratio)),
           "cleaned_length_ratio_max": float(np.max(len_ratio)), "cleaned_length_ratio_min": float(np.min(len_ratio)),
           "note": "reference: jps3d jump point search + tolerance comparator on a binary heap; here: A*, exact empty-grid heuristic, total order. "
                   "Equal raw-path cost; the cleaned vertex list (removeLinePts/removeCornerPts) differs when another equal-cost path is found."}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ref_frontend_path_compare.json"), "w") as f:
        json.dump(out, f, indent=1)
    assert 0.1 < frac <= 1.0 and 0.9 < np.mean(len_ratio) < 1.1


@pytest.mark.parametrize("seed,start,goal", [(7, (0.8, 0.9, 1.0), (9.1, 9.2, 1.1)), (8, (9.0, 0.8, 0.4), (1.0, 9.3, 1.7))])
def test_jps3d_raw_cost_is_the_dijkstra_optimum(ref, seed, start, goal):
    """The reference's raw path cost against SciPy's Dijkstra on the reference's own occupancy grid (start / goal surroundings
    freed as solveJPS3D does): the optimum that this repository's search is checked against in test_frontend.py."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra

    cloud, _ = frontend.forest_cloud(seed, size=(10.0, 10.0, 2.0), density=0.15)
    cloud = cloud.astype(np.float32).astype(np.float64)
    res, infl = 0.25, 0.25
    cells, center = (44, 44, 8), np.array([5.0, 5.0, 1.0])
    start, goal = np.array(start), np.array(goal)
    m = ref.Map(cloud, cells, res, center, 0.0, 2.0, infl)
    p, cost, _ = m.plan(start, goal, True)
    assert p is not None
    occ = (m.occupancy() > 0).transpose(2, 1, 0).copy()   # [x][y][z]
    nx, ny, nz = occ.shape
    n_free = int(round(infl / res + 0.5))                 # setFreeVoxelAndSurroundings (map_util.h:250-265)

    def cell(q):
        return np.round((q - m.origin) / res - 0.5).astype(int)

    for c in (cell(start), cell(goal)):
        occ[max(c[0] - n_free, 0):c[0] + n_free + 1, max(c[1] - n_free, 0):c[1] + n_free + 1, max(c[2] - n_free, 0):c[2] + n_free + 1] = False
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
    assert cost == pytest.approx(dist * res, rel=1e-9)
    m.close()


def test_start_and_goal_inside_the_inflated_obstacles(ref):
    """setFreeVoxelAndSurroundings(center, const float d) frees round(d / res + 0.5) cells around start and goal (map_util.h:248-263;
    inflation 0.3 m at 0.2 m cells: TWO cells, not the one cell of readMap's own inflation): queries that start or end inside the
    inflated hull of a tree — where that number decides whether a path exists at all — exist for jps3d iff they exist here and cost
    the same.  (Rounds 1-2 freed floor(inflation / res) cells; found when the reference's own map code could be run.)"""
    cloud, centres = frontend.forest_cloud(3)
    cloud = cloud.astype(np.float32).astype(np.float64)
    res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
    cells, center = (110, 110, 15), np.array([10.0, 10.0, 1.5])
    rng = np.random.default_rng(11)
    n = 96
    ang = rng.uniform(0, 2 * np.pi, n)
    near = centres[rng.integers(0, len(centres), n)] + np.column_stack([np.cos(ang), np.sin(ang)]) * rng.uniform(0.40, 0.75, (n, 1))
    starts = np.column_stack([near, rng.uniform(0.8, 2.2, n)])
    goals = np.column_stack([rng.uniform(2, 18, n), rng.uniform(2, 18, n), rng.uniform(0.8, 2.2, n)])
    starts[n // 2:], goals[n // 2:] = goals[n // 2:].copy(), starts[n // 2:].copy()    # half of them END inside the hull
    hp, hn, _ = frontend.plan_batch(cloud, cells, res, center, zg, zmax, infl, starts, goals, max_points=256)
    m = ref.Map(cloud, cells, res, center, zg, zmax, infl)
    occ = m.occupancy()
    inside = found = 0
    for i in range(n):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        c = np.round((np.minimum(starts[i], goals[i]) * 0 + (starts[i] if i < n // 2 else goals[i]) - m.origin) / res - 0.5).astype(int)
        inside += int(occ[c[2], c[1], c[0]] > 0)
        assert (p is None) == (hn[i] == 0), (i, p is None, hn[i])
        if p is not None:
            found += 1
            np.testing.assert_allclose(hp[i, 0], p[0], atol=1e-12)
            np.testing.assert_allclose(hp[i, hn[i] - 1], p[-1], atol=1e-12)
    m.close()
    assert inside >= 20 and found >= 40, (inside, found)


# ---- jump point search in jps3d's own order (fhfront::plan_path_jps): FASTER's exact path ------------------------------------------
def test_jps_neighbour_tables_equal_jps3d(ref):
    """The tables plan_path_jps GENERATES from geometric rules (natural neighbours, cells to test, directions to add per move) equal
    the ones jps3d's JPS3DNeib constructor writes down case by case (graph_search.cpp:573-937), entry by entry in the same order —
    the order decides in which sequence successors reach the open list, hence which of several equal-cost paths is found."""
    a, b = ref.jps3d_tables(), frontend.jps_tables()
    count = {0: (26, 0), 1: (1, 8), 2: (3, 12), 3: (7, 12)}
    for idx in range(27):
        d = (idx % 3 - 1, (idx // 3) % 3 - 1, idx // 9 - 1)
        nn, nf = count[abs(d[0]) + abs(d[1]) + abs(d[2])]
        assert np.array_equal(a[0][idx, :, :nn], b[0][idx, :, :nn]), ("ns", d)
        assert np.array_equal(a[1][idx, :, :nf], b[1][idx, :, :nf]), ("f1", d)
        assert np.array_equal(a[2][idx, :, :nf], b[2][idx, :, :nf]), ("f2", d)


@pytest.mark.parametrize("seed,res,inflation,n", [(3, 0.2, 0.3, 400), (4, 0.25, 0.25, 200), (5, 0.15, 0.3, 120)])
def test_jps_vertex_lists_equal_jps3d(ref, seed, res, inflation, n):
    """plan_path_jps against the reference's compiled graph_search.cpp / jps_planner.cpp driven as solveJPS3D drives them: the same
    raw cost and the same cleaned vertex list, vertex for vertex and bit for bit, for every query (jump point pruning, the
    successor order, the tolerance comparator f within 1e-6 => smaller g first, the sift discipline of the binary heap, the path
    clean-up).  This is the path FASTER feeds its decomposition."""
    cloud, cells, center, starts, goals = forest(seed, n, res=res, inflation=inflation)
    zg, zmax = 0.0, 3.0
    m = ref.Map(cloud, cells, res, center, zg, zmax, inflation)
    found = 0
    for i in range(n):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        hp, hcost, hex_ = frontend.plan_jps(cloud, cells, res, center, zg, zmax, inflation, starts[i], goals[i])
        assert (p is None) == (hp is None), i
        if p is None:
            continue
        found += 1
        assert len(p) == len(hp) and np.array_equal(p, hp), (i, len(p), len(hp))
        assert hcost == pytest.approx(cost, abs=1e-9)
    m.close()
    assert found >= 0.95 * n


def test_jps_vertex_lists_equal_jps3d_inside_the_inflated_obstacles(ref):
    """The same for queries that start or end inside the inflated hull of a tree (the cells freed around start and goal matter)."""
    cloud, centres = frontend.forest_cloud(3)
    cloud = cloud.astype(np.float32).astype(np.float64)
    res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
    cells, center = (110, 110, 15), np.array([10.0, 10.0, 1.5])
    rng = np.random.default_rng(12)
    n = 96
    ang = rng.uniform(0, 2 * np.pi, n)
    near = centres[rng.integers(0, len(centres), n)] + np.column_stack([np.cos(ang), np.sin(ang)]) * rng.uniform(0.40, 0.75, (n, 1))
    starts = np.column_stack([near, rng.uniform(0.8, 2.2, n)])
    goals = np.column_stack([rng.uniform(2, 18, n), rng.uniform(2, 18, n), rng.uniform(0.8, 2.2, n)])
    starts[n // 2:], goals[n // 2:] = goals[n // 2:].copy(), starts[n // 2:].copy()
    m = ref.Map(cloud, cells, res, center, zg, zmax, infl)
    found = 0
    for i in range(n):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        hp, hcost, _ = frontend.plan_jps(cloud, cells, res, center, zg, zmax, infl, starts[i], goals[i])
        assert (p is None) == (hp is None), i
        if p is not None:
            found += 1
            assert len(p) == len(hp) and np.array_equal(p, hp), i
    m.close()
    assert found >= 40


@pytest.mark.parametrize("seed", [3, 4])
def test_jps_vertex_lists_equal_jps3d_on_the_c5_queries(ref, seed):
    """The start/goal pairs of config C5's generator.  Seed 4, query 112 is the case that pinned the ORDER of the ray test's arithmetic
    (rayTrace, map_util.h:349-370: pt = pt1 + (diff * s) * n): the samples of a diagonal ray sit on cell corners, diff * (s * n)
    rounds to the other cell once in a few thousand rays, removeCornerPts then keeps another vertex."""
    cloud, cells, center, starts, goals = frontend.forest_queries(160, seed)
    cloud = cloud.astype(np.float32).astype(np.float64)
    res, zg, zmax, infl = 0.2, 0.0, 3.0, 0.3
    m = ref.Map(cloud, cells, res, center, zg, zmax, infl)
    found = 0
    for i in range(len(starts)):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        hp, hcost, _ = frontend.plan_jps(cloud, cells, res, center, zg, zmax, infl, starts[i], goals[i])
        assert (p is None) == (hp is None), i
        if p is not None:
            found += 1
            assert len(p) == len(hp) and np.array_equal(p, hp), i
    m.close()
    assert found >= 150


@pytest.mark.parametrize("inflation", [0.25, 0.6])
def test_jps_vertex_lists_equal_jps3d_in_a_cubic_map_with_blobs(ref, inflation):
    """A map as high as it is wide (40^3 cells) with random blobs: space-diagonal jumps, goals above and below the start, and (0.6 m
    inflation at 0.25 m cells) large freed cubes that overlap the obstacles — plan_path_jps against the reference's compiled jps3d."""
    rng = np.random.default_rng(11)
    res, zg, zmax = 0.25, 0.0, 10.0
    cells, center = (40, 40, 40), np.array([5.0, 5.0, 5.0])
    blobs = rng.uniform(0.5, 9.5, (60, 3))
    cloud = (blobs[:, None, :] + rng.normal(0, 0.25, (60, 40, 3))).reshape(-1, 3).astype(np.float32).astype(np.float64)
    n = 160
    starts, goals = rng.uniform(0.3, 9.7, (n, 3)), rng.uniform(0.3, 9.7, (n, 3))
    m = ref.Map(cloud, cells, res, center, zg, zmax, inflation)
    found = 0
    for i in range(n):
        p, cost, _ = m.plan(starts[i], goals[i], True)
        hp, hcost, _ = frontend.plan_jps(cloud, cells, res, center, zg, zmax, inflation, starts[i], goals[i])
        assert (p is None) == (hp is None), i
        if p is not None:
            found += 1
            assert len(p) == len(hp) and np.array_equal(p, hp), i
    m.close()
    assert found >= 120
