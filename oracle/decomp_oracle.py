"""numpy restatement of FASTER's convex decomposition around a path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: DecompUtil is header-only C++ on Eigen (absent here) and its tests only write SVGs
(thirdparty/DecompROS/DecompUtil/test/*.cpp; no assertions), so there is nothing to diff against.
This module is an independent implementation, written from the geometry, of what
JPS_Manager::cvxEllipsoidDecomp (faster/src/jps_manager.cpp:80-127) computes through
EllipsoidDecomp3D::dilate / LineSegment3D::dilate (DecompUtil line_segment.h:34-39, :57-98, :156-252,
decomp_base.h:83-115, ellipsoid.h:24-73, polyhedron.h:131-152), used to check faster_amd/host/corridor_frontend.hpp.
"""
import numpy as np

EPS = 1e-10  # DecompUtil epsilon_


def rot_x_to(v):
    pitch = np.arctan2(-v[2], np.hypot(v[0], v[1]))
    yaw = np.arctan2(v[1], v[0])
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    return np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1.0]]) @ np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])


def rot_x(roll):
    c, s = np.cos(roll), np.sin(roll)
    return np.array([[1.0, 0, 0], [0, c, -s], [0, s, c]])


def local_bbox_planes(p1, p2, bbox):
    d = (p2 - p1) / np.linalg.norm(p2 - p1)
    h = np.array([d[1], -d[0], 0.0])
    if np.linalg.norm(h) == 0:
        h = np.array([-1.0, 0, 0])
    h = h / np.linalg.norm(h)
    v = np.cross(d, h)
    return [(p1 + h * bbox[1], h), (p1 - h * bbox[1], -h), (p2 + d * bbox[0], d), (p1 - d * bbox[0], -d),
            (p1 + v * bbox[2], v), (p1 - v * bbox[2], -v)]


def ell_dist(R, axes, c, pts):
    loc = (pts - c) @ R            # rows: R^T (q - c)
    return np.linalg.norm(loc / axes, axis=1)


def decompose_segment(p1, p2, cloud, bbox=(2.0, 2.0, 1.0), inflate=0.0):
    """Returns (planes [(point, unit normal)...], (R, axes, centre)) for one path segment."""
    p1, p2, cloud = np.asarray(p1, float), np.asarray(p2, float), np.asarray(cloud, float).reshape(-1, 3)
    box = local_bbox_planes(p1, p2, bbox)
    keep = np.ones(len(cloud), bool)
    for q, n in box:
        keep &= (cloud - q) @ n <= EPS
    obs = cloud[keep].copy()
    f = np.linalg.norm(p1 - p2) / 2
    Ri = rot_x_to(p2 - p1)
    c = (p1 + p2) / 2
    axes = np.array([f, f, f])
    if len(obs):  # FASTER's obstacle inflation towards the centre, in the ellipsoid frame
        loc = (obs - c) @ Ri
        obs = (loc - np.sign(loc) * inflate) @ Ri.T + c
    first = obs[ell_dist(Ri, axes, c, obs) <= 1] if len(obs) else obs
    inside = first
    Rf = Ri
    while len(inside):
        pw = inside[np.argmin(ell_dist(Rf, np.array([axes[0], axes[1], axes[1]]), c, inside))]  # closest in the CURRENT ellipsoid
        loc = Ri.T @ (pw - c)
        Rf = Ri @ rot_x(np.arctan2(loc[2], loc[1]))
        loc = Rf.T @ (pw - c)
        if loc[0] < axes[0]:
            axes[1] = abs(loc[1]) / np.sqrt(1 - (loc[0] / axes[0]) ** 2)
        cur = np.array([axes[0], axes[1], axes[1]])
        inside = inside[1 - ell_dist(Rf, cur, c, inside) > EPS]
    inside = first[ell_dist(Rf, axes, c, first) <= 1] if len(first) else first
    while len(inside):
        pw = inside[np.argmin(ell_dist(Rf, axes, c, inside))]
        loc = Rf.T @ (pw - c)
        dd = 1 - (loc[0] / axes[0]) ** 2 - (loc[1] / axes[1]) ** 2
        if dd > EPS:
            axes[2] = abs(loc[2]) / np.sqrt(dd)
        inside = inside[1 - ell_dist(Rf, axes, c, inside) > EPS]
    planes = []
    remain = obs
    while len(remain):
        cp = remain[np.argmin(ell_dist(Rf, axes, c, remain))]
        g = Rf @ ((Rf.T @ (cp - c)) / axes**2)
        n = g / np.linalg.norm(g)
        planes.append((cp, n))
        remain = remain[(remain - cp) @ n < 0]
    planes += box
    return planes, (Rf, axes.copy(), c)


def constraint(inside_pt, planes):
    A, b = [], []
    for q, n in planes:
        off = q @ n
        if n @ inside_pt - off > 0:
            n, off = -n, -off
        A.append(n)
        b.append(off)
    return np.array(A), np.array(b)


def decompose_path(path, cloud, drone_radius, z_ground, bbox=(2.0, 2.0, 1.0)):
    """cvxEllipsoidDecomp: list of (A, b) per path segment, ground plane -z <= -z_ground appended."""
    out = []
    path = np.asarray(path, float)
    for i in range(len(path) - 1):
        planes, _ = decompose_segment(path[i], path[i + 1], cloud, bbox, drone_radius)
        A, b = constraint((path[i] + path[i + 1]) / 2, planes)
        out.append((np.vstack([A, [0, 0, -1.0]]), np.append(b, -z_ground)))
    return out
