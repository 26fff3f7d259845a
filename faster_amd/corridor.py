"""Synthetic corridor / problem generator; numpy only, seeded, vectorised over the batch.

The reference builds its corridors with JPS + DecompUtil's ellipsoid decomposition
(faster/src/jps_manager.cpp:80-127; DecompUtil line_segment.h:57-98 adds a local bounding box around
each path segment, jps_manager.cpp:113-124 appends the ground plane).  That front-end is out of scope
(SURVEY.md §8(f) N1); this module emulates its output distribution:

  * a random polyline of P segments, length U(1.5, 4) m, heading change <= 60 deg, z in [0.5, 2.5];
  * polytope p = oriented box around segment p (2 m ahead/behind, 2 m sideways, 1 m vertically: the
    local bbox of jps_manager.cpp:100) + K~U{2..8} random cutting planes that keep the whole segment
    inside with a margin U(0.5, 2) m (emulating obstacle hyperplanes) + the ground plane -z <= 0;
    all normals are unit vectors, consecutive polytopes overlap around the shared vertex;
  * x0: position = first vertex, velocity mostly along the first segment, small acceleration;
    xf: position = last vertex, zero velocity/acceleration (as E and M_ in faster.cpp:376-377,394);
  * parameters from faster/param/faster.yaml:5,23-25 (dc 0.01, v/a/j max 5/5/8) and the first-replan
    factor window [1, 10] step 1 (faster.cpp:57,68).

BASELINE configs:  C2 = safe_batch(1024, seed=1), C3 = whole_batch(4096, seed=2, P<=4),
C4 = whole_batch(32768, seed=3, P<=6) + pair glue, C5 fallback = whole_batch(65536, seed=5, N=15, P<=8).
"""
import numpy as np

from . import abi

_K_MAX = 8
_F_MAX = 6 + _K_MAX + 1


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def make_batch(n, n_seg, p_choices, force_final, seed, v_max=5.0, a_max=5.0, j_max=8.0, dc=0.01,
               f_init=1.0, f_final=10.0, f_inc=1.0, speed=2.5, lateral=0.5, acc0=0.5):
    """Returns (problems[n] of abi.problem_dtype, faces[...] of abi.face_dtype, verts[n, Pmax+1, 3])."""
    rng = np.random.default_rng(seed)
    p_choices = np.asarray(p_choices, dtype=np.int64)
    pmax = int(p_choices.max())
    P = rng.choice(p_choices, size=n)

    verts = np.zeros((n, pmax + 1, 3))
    verts[:, 0, 0:2] = rng.uniform(-10, 10, size=(n, 2))
    verts[:, 0, 2] = rng.uniform(0.8, 2.2, size=n)
    heading = rng.uniform(0, 2 * np.pi, size=n)
    for p in range(pmax):
        if p > 0:
            heading = heading + rng.uniform(-np.pi / 3, np.pi / 3, size=n)
        length = rng.uniform(1.5, 4.0, size=n)
        z = np.clip(verts[:, p, 2] + rng.uniform(-0.5, 0.5, size=n), 0.5, 2.5)
        dz = z - verts[:, p, 2]
        horiz = np.sqrt(np.maximum(length**2 - dz**2, 1.0))
        verts[:, p + 1, 0] = verts[:, p, 0] + horiz * np.cos(heading)
        verts[:, p + 1, 1] = verts[:, p, 1] + horiz * np.sin(heading)
        verts[:, p + 1, 2] = z

    fa = np.zeros((n, pmax, _F_MAX, 4))
    valid = np.zeros((n, pmax, _F_MAX), dtype=bool)
    ez = np.array([0.0, 0.0, 1.0])
    for p in range(pmax):
        a, b = verts[:, p], verts[:, p + 1]
        u = _unit(b - a)
        w1 = _unit(np.cross(u, ez))
        w2 = np.cross(u, w1)
        box = [(u, 2.0, b), (-u, 2.0, a), (w1, 2.0, a), (-w1, 2.0, a), (w2, 1.0, a), (-w2, 1.0, a)]
        for k, (nrm, off, ref) in enumerate(box):
            fa[:, p, k, 0:3] = nrm
            fa[:, p, k, 3] = np.einsum("ij,ij->i", nrm, ref) + off
            valid[:, p, k] = True
        K = rng.integers(2, _K_MAX + 1, size=n)
        nk = _unit(rng.normal(size=(n, _K_MAX, 3)))
        margin = rng.uniform(0.5, 2.0, size=(n, _K_MAX))
        reach = np.maximum(np.einsum("nkj,nj->nk", nk, a), np.einsum("nkj,nj->nk", nk, b))
        fa[:, p, 6:6 + _K_MAX, 0:3] = nk
        fa[:, p, 6:6 + _K_MAX, 3] = reach + margin
        valid[:, p, 6:6 + _K_MAX] = np.arange(_K_MAX)[None, :] < K[:, None]
        fa[:, p, _F_MAX - 1] = np.array([0.0, 0.0, -1.0, 0.0])  # ground plane -z <= 0 (jps_manager.cpp:118-122)
        valid[:, p, _F_MAX - 1] = True
        valid[:, p] &= (p < P)[:, None]

    counts = valid.sum(axis=2)  # [n, pmax]
    face_off = np.zeros((n, abi.FH_MAX_POLY + 1), dtype=np.int32)
    face_off[:, 1:pmax + 1] = np.cumsum(counts, axis=1)
    for p in range(pmax + 1, abi.FH_MAX_POLY + 1):
        face_off[:, p] = face_off[:, pmax]
    per_problem = counts.sum(axis=1)
    face_begin = np.concatenate([[0], np.cumsum(per_problem)[:-1]]).astype(np.int32)

    flat = fa[valid]  # row order: problem, polytope, face
    faces = np.zeros(flat.shape[0], dtype=abi.face_dtype)
    faces["a"] = flat[:, 0:3]
    faces["b"] = flat[:, 3]

    pr = abi.make_problems(n)
    pr["n_seg"] = n_seg
    pr["n_poly"] = P
    pr["force_final_pos"] = 1 if force_final else 0
    pr["face_begin"] = face_begin
    pr["face_off"] = face_off
    pr["dc"] = dc
    pr["v_max"], pr["a_max"], pr["j_max"] = v_max, a_max, j_max
    pr["f_init"], pr["f_final"], pr["f_inc"] = f_init, f_final, f_inc
    u0 = _unit(verts[:, 1] - verts[:, 0])
    vel = u0 * rng.uniform(0, speed, size=(n, 1)) + rng.uniform(-lateral, lateral, size=(n, 3))
    acc = rng.uniform(-acc0, acc0, size=(n, 3))
    pr["x0"][:, 0:3] = verts[:, 0]
    pr["x0"][:, 3:6] = vel
    pr["x0"][:, 6:9] = acc
    pr["xf"][:, 0:3] = verts[np.arange(n), P]
    return pr, faces, verts


def whole_batch(n, seed, n_seg=10, p_choices=(2, 3, 4), **kw):
    """Whole-trajectory problems (final position forced; sg_whole_ in faster.cpp:52-60)."""
    return make_batch(n, n_seg, p_choices, True, seed, **kw)


def safe_batch(n, seed, n_seg=6, p_choices=(1,), **kw):
    """Safe-trajectory problems (final position free; sg_safe_ in faster.cpp:63-71). With one polytope the
    binaries are forced and the problem is a pure QP (BASELINE config 2)."""
    pr, faces, verts = make_batch(n, n_seg, p_choices, False, seed, **kw)
    return pr, faces, verts


def safe_templates(whole):
    """Safe problems paired with whole problems (BASELINE config 4): same N, bounds, dc and factor window,
    force_final_pos = 0, goal position = the whole goal (used only for dt, faster.cpp:522).  x0 and the
    corridor (n_poly, face_begin, face_off) are filled on the device by fh_pair_glue_device."""
    s = whole.copy()
    s["force_final_pos"] = 0
    s["n_poly"] = 0
    s["face_off"] = 0
    s["x0"] = 0
    return s


def fixture_problem(fx, n_seg, vaj, force_final, polys, x0, xf, f_init=1.0, f_final=10.0, f_inc=1.0, dc=0.01):
    """Problem on the reference's hard-coded corridor (tests/golden/fixture_corridor.json)."""
    faces, off = abi.pack_faces([(fx["polytopes"][p]["A"], fx["polytopes"][p]["b"]) for p in polys])
    pr = abi.make_problems(1)
    pr["n_seg"] = n_seg
    pr["n_poly"] = len(polys)
    pr["force_final_pos"] = 1 if force_final else 0
    pr["face_off"][0, : len(off)] = off
    pr["face_off"][0, len(off):] = off[-1]
    pr["dc"] = dc
    pr["v_max"], pr["a_max"], pr["j_max"] = vaj
    pr["f_init"], pr["f_final"], pr["f_inc"] = f_init, f_final, f_inc
    pr["x0"][0] = x0
    pr["xf"][0] = xf
    return pr, faces


def concat(batches):
    """Concatenate (problems, faces) batches, rebasing face_begin."""
    prs, fcs, base = [], [], 0
    for pr, fc in batches:
        pr = pr.copy()
        pr["face_begin"] += base
        base += fc.shape[0]
        prs.append(pr)
        fcs.append(fc)
    return np.concatenate(prs), np.concatenate(fcs)
