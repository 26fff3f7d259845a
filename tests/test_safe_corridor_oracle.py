"""CPU checks of the restatement the device's safe corridor is compared with (oracle/pair_glue.py: safe_path, unknown_voxels;
faster/src/faster.cpp:446-524): properties that follow from the reference's construction, and the explicit-cloud decomposition the
GPU test uses (unknown voxels listed before the occupied points) against a decomposition that sees the unknown voxels only."""
import numpy as np

from faster_amd import frontend
from oracle import pair_glue


def test_safe_path_is_cut_where_known_space_ends_and_starts_at_r():
    rng = np.random.default_rng(4)
    for _ in range(200):
        A = rng.uniform(-2, 2, 3)
        steps = rng.uniform(-1.0, 1.0, (8, 3)) + np.array([1.2, 0.3, 0.0])
        path = np.vstack([A, A + np.cumsum(steps, axis=0)])
        R = A + rng.uniform(-0.5, 0.5, 3)
        r_known, drone_r, mps = 4.0, 0.3, int(rng.integers(1, 5))
        sp = pair_glue.safe_path(path, A, R, r_known, drone_r, mps)
        assert np.array_equal(sp[0], R) and 2 <= len(sp) <= mps + 1
        leaves = np.linalg.norm(path - A, axis=1).max() > r_known
        if leaves and len(sp) < mps + 1:   # the end of the path is the cut: on the path, drone_radius (along the path) before the sphere
            end = sp[-1]
            assert np.linalg.norm(end - A) <= r_known + 1e-5
            d = [np.linalg.norm(np.cross(path[i + 1] - path[i], end - path[i])) / np.linalg.norm(path[i + 1] - path[i]) for i in range(len(path) - 1)]
            assert min(d) < 1e-5
        if not leaves:                      # known space all along: the path as it was (first vertex replaced, legs limited)
            assert np.array_equal(sp[1:], path[1:mps + 1])


def test_safe_path_when_the_start_is_already_at_the_boundary():
    A = np.zeros(3)
    path = np.array([[3.9, 0, 0], [5.0, 0, 0], [6.0, 0, 0]])   # first vertex 0.1 m from unknown space, drone_radius 0.3
    sp = pair_glue.safe_path(path, A, path[0], 4.0, 0.3, 3)
    assert len(sp) == 2 and np.allclose(sp[1], [3.91, 0, 0])   # the reference's 1 cm stub (faster.cpp:813-826)


def test_unknown_voxels_are_the_far_cells_in_cloud_order():
    origin, res, dims, A, r = np.array([-1.0, -2.0, 0.0]), 0.5, np.array([9, 7, 4]), np.array([1.0, 0.5, 1.0]), 1.6
    u = pair_glue.unknown_voxels(origin, res, dims, A, r)
    want = []
    for iz in range(dims[2]):
        for iy in range(dims[1]):
            for ix in range(dims[0]):
                c = (np.array([ix, iy, iz]) + 0.5) * res + origin
                if ((c - A) ** 2).sum() > r * r:
                    want.append(c)
    assert np.array_equal(u, np.array(want)) and 0 < len(u) < dims.prod()


def test_unknown_voxels_bound_the_corridor():
    """A path in empty space: without unknown voxels the polytope is the local bounding box; with them no vertex of it lies beyond
    the sphere of known space by more than a cell."""
    A = np.array([5.0, 5.0, 1.5])
    path = np.array([[5.0, 5.0, 1.5], [6.4, 5.3, 1.5]])
    origin, res, dims = np.array([0.0, 0.0, 0.0]), 0.2, np.array([50, 50, 15])
    free, _ = frontend.decompose(path, np.zeros((0, 3)), drone_radius=0.05, z_ground=0.0)
    unk = pair_glue.unknown_voxels(origin, res, dims, A, 2.5)
    bounded, _ = frontend.decompose(path, unk, drone_radius=0.05, z_ground=0.0)
    assert len(bounded[0][1]) > len(free[0][1])
    Ab, bb = bounded[0]
    for q in unk[::7]:
        assert np.any(Ab @ q - bb > -1e-9)   # every unknown voxel is outside (or on) the polytope
    assert np.all(Ab @ path[0] - bb < 0) and np.all(Ab @ path[1] - bb < 0)


def test_host_paths_clipped_to_the_sphere_equal_the_restatement():
    """ff_set_sphere (JPS_in of Faster::replan, faster.cpp:370-382) in the C++ front-end against oracle/pair_glue.clip_to_sphere applied
    to the unclipped paths: the same vertices bit for bit (the crossing point in the reference's mixed single/double arithmetic), and
    every clipped path ends on the sphere of radius min(|goal - start| - 0.001, Ra)."""
    cloud, cells, center, starts, goals = frontend.forest_queries(192, 9)
    res, zmax, infl, Ra = 0.2, 3.0, 0.3, 4.0
    frontend.set_search("jps")
    try:
        plain = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=64)
        frontend.set_sphere(Ra)
        clipped = frontend.plan_batch(cloud, cells, res, center, 0.0, zmax, infl, starts, goals, max_points=64)
    finally:
        frontend.set_sphere(0.0)
        frontend.set_search("astar")
    n_cut = 0
    for i in range(len(starts)):
        if plain[1][i] <= 0:
            assert clipped[1][i] == plain[1][i]
            continue
        want = pair_glue.clip_to_sphere(plain[0][i, :plain[1][i]], Ra)
        got = clipped[0][i, :clipped[1][i]]
        assert got.shape == want.shape and np.array_equal(got, want), i
        if len(want) != plain[1][i] or not np.array_equal(want[-1], plain[0][i, plain[1][i] - 1]):
            n_cut += 1
            ra = min(np.linalg.norm(plain[0][i, plain[1][i] - 1] - plain[0][i, 0]) - 0.001, Ra)
            assert abs(np.linalg.norm(got[-1] - got[0]) - ra) < 1e-4   # (the crossing point is a single-precision result)
    assert n_cut > 100
