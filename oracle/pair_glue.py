"""Host restatement of fh_pair_glue_device (whole -> safe hand-off for synthetic pairs).  TEST INFRASTRUCTURE ONLY.

Mirrors the data dependency of Faster::replan (faster/src/faster.cpp:456-475: R = X_whole[k]; :506-524: the safe
solver starts from R) for the synthetic pairing of SURVEY.md §8(d) C4.  Uses oracle.sample (fillX semantics)."""
import numpy as np

from faster_amd import abi
from . import oracle


def find_index_h(X, x0_pos, r_known, drone_radius, delta_h):
    """Faster::findIndexH (faster/src/faster.cpp:218-251) with the unknown space of fh_set_pair_rule: every 10th sample; the first one
    whose nearest unknown point (modelled: anything farther than r_known from x0) is closer than drone_radius.  -> (indexH, needed)"""
    for i in range(0, X.shape[0], 10):
        if r_known - np.linalg.norm(X[i]["pos"] - x0_pos) < drone_radius:
            return int(delta_h * i), True
    return X.shape[0] - 1, False


def find_index_r(X, index_h, delta_a, a_max):
    """Faster::findIndexR (faster.cpp:173-216), literally: x and y only."""
    pos_h = X[index_h]["pos"][:2]
    for i in range(0, index_h + 1):
        vel, pos = X[i]["vel"][:2], X[i]["pos"][:2]
        braking = np.sign(vel * (pos_h - pos)) * vel ** 2 / (2 * delta_a * a_max)
        if np.any(braking > np.abs(pos_h - pos)):
            return i
    return index_h


def glue(whole, wres, faces, safe_templates, r_frac=0.5, shrink=0.2, max_safe_poly=3, r_margin=-1.0, rule=None):
    """rule: None (R at the fraction r_frac of the samples) or dict(r_known, drone_radius, delta_h, delta_a): FASTER's own choice of R."""
    keep_r = r_margin >= 0
    safe = safe_templates.copy()
    sfaces = np.zeros_like(faces)
    for i in range(len(whole)):
        pw, rw = whole[i], wres[i]
        if not rw["solved"]:
            safe["n_seg"][i] = 0
            continue
        X = oracle.sample(pw, rw)
        size = X.shape[0]
        k = min(max(int(r_frac * size), 0), size - 1)
        if rule is not None:
            index_h, needed = find_index_h(X, pw["x0"][:3], rule["r_known"], rule["drone_radius"], rule.get("delta_h", 1.0))
            if not needed:          # needToComputeSafePath == false (faster.cpp:462-466): no safe trajectory
                safe["n_seg"][i] = 0
                continue
            k = find_index_r(X, min(index_h, size - 1), rule.get("delta_a", 0.5), float(pw["a_max"]))
        R = X[k]
        safe["x0"][i, 0:3], safe["x0"][i, 3:6], safe["x0"][i, 6:9] = R["pos"], R["vel"], R["accel"]
        P = int(pw["n_poly"])
        fb = int(pw["face_begin"])
        start, best, found = 0, np.inf, False
        for p in range(P):
            f0, f1 = fb + pw["face_off"][p], fb + pw["face_off"][p + 1]
            A, b = faces["a"][f0:f1], faces["b"][f0:f1]
            nr = np.sqrt((A * A).sum(axis=1))
            worst = np.max(A @ R["pos"] - (b - (0.0 if keep_r else shrink) * nr)) if f1 > f0 else -np.inf
            if worst <= (1e-7 if keep_r else 0.0):
                start, found = p, True
                break
            if worst < best:
                best, start = worst, p
        cnt = min(P - start, max_safe_poly) if P else 0
        safe["n_poly"][i] = cnt
        safe["face_begin"][i] = fb
        o = 0
        off = [0]
        for p in range(cnt):
            f0, f1 = fb + pw["face_off"][start + p], fb + pw["face_off"][start + p + 1]
            m = f1 - f0
            sfaces["a"][fb + o: fb + o + m] = faces["a"][f0:f1]
            nr = np.sqrt((faces["a"][f0:f1] ** 2).sum(axis=1))
            bb = faces["b"][f0:f1] - shrink * nr
            if keep_r and p == 0:  # no face of the polytope that holds R is pulled closer to R than r_margin
                ar = faces["a"][f0:f1] @ R["pos"]
                bb = np.maximum(bb, np.maximum(np.minimum(faces["b"][f0:f1], ar + r_margin * nr), ar + 1e-6 * nr))
            sfaces["b"][fb + o: fb + o + m] = bb
            o += m
            off.append(o)
        off += [o] * (abi.FH_MAX_POLY + 1 - len(off))
        safe["face_off"][i] = off
    return safe, sfaces


def append_plans(whole, wres, safe, sres, r_frac=0.5, rule=None):
    """Faster::appendToPlan (faster/src/faster.cpp:606-648) for independent pairs whose plan holds only the start (k_end_whole = 0):
    plan = X_whole[0 .. k_safe] followed by X_safe (:627-640); nothing is committed when the whole solve failed (:427-431) or a safe
    trajectory was needed and not found (:529-533); without unknown space on the way (:462-466) the plan is the whole trajectory.
    -> (list of state arrays (empty: nothing committed), k_safe per pair (-1: nothing committed))"""
    plans, ks = [], []
    for i in range(len(whole)):
        pw, rw = whole[i], wres[i]
        if not rw["solved"]:
            plans.append(None); ks.append(-1)
            continue
        X = oracle.sample(pw, rw)
        size = X.shape[0]
        k, needed = min(max(int(r_frac * size), 0), size - 1), True
        if rule is not None:
            index_h, needed = find_index_h(X, pw["x0"][:3], rule["r_known"], rule["drone_radius"], rule.get("delta_h", 1.0))
            k = find_index_r(X, min(index_h, size - 1), rule.get("delta_a", 0.5), float(pw["a_max"])) if needed else size - 1
        if needed and not (sres[i]["solved"] and safe[i]["n_seg"] >= 1):
            plans.append(None); ks.append(-1)
            continue
        parts = [X[:k + 1]]
        if needed:
            parts.append(oracle.sample(safe[i], sres[i]))
        plans.append(np.concatenate(parts))
        ks.append(k)
    return plans, np.array(ks, dtype=np.int32)


# ---- the safe corridor decomposed around R (faster/src/faster.cpp:446-524): restatement of faster_amd/csrc/fh_safe.hip.hpp ----------
def _sphere_crossing(a_in, b_in, r, c):
    """Point where the segment a -> b leaves the sphere (centre c, radius r): getIntersectionWithSphere (utils.cpp:713-776) with its
    arithmetic — single precision except where the language promotes (pow(float, 2), `- r * r`)."""
    f = np.float32
    def solve(A, B):
        x1, y1, z1, x2, y2, z2 = f(A[0]), f(A[1]), f(A[2]), f(B[0]), f(B[1]), f(B[2])
        x3, y3, z3 = f(c[0]), f(c[1]), f(c[2])
        dx, dy, dz = x2 - x1, y2 - y1, z2 - z1
        a = f(float(dx) * float(dx) + float(dy) * float(dy) + float(dz) * float(dz))   # pow(float, 2) is a double: summed in double, rounded once
        b = f(2) * (dx * (x1 - x3) + dy * (y1 - y3) + dz * (z1 - z3))
        cf = x3 * x3 + y3 * y3 + z3 * z3 + x1 * x1 + y1 * y1 + z1 * z1 - f(2) * (x3 * x1 + y3 * y1 + z3 * z1)
        cc = f(float(cf) - r * r)                                                       # `- r * r`: a double subtraction
        disc = b * b - f(4) * a * cc
        with np.errstate(invalid="ignore", divide="ignore"):
            t = (-b + np.sqrt(disc)) / (f(2) * a)
            p = np.array([float(x1 + dx * t), float(y1 + dy * t), float(z1 + dz * t)])
        return disc, p
    disc, p = solve(a_in, b_in)
    if disc <= 0:
        return solve(c, a_in)[1]
    return p


def _sphere_exit(path, r, center):
    """First point of the path on the sphere (utils.cpp:782-870) -> (point, last index inside, none outside)"""
    index = -1
    for i, v in enumerate(path):
        if np.sqrt(((v - center) ** 2).sum()) > r:
            index = i
            break
    if index == -1:
        return _sphere_crossing(center, path[-1], r, center), len(path) - 1, True
    if index == 0:
        return path[0].copy(), 1, False
    return _sphere_crossing(path[index - 1], path[index], r, center), index - 1, False


def clip_to_sphere(path, Ra):
    """JPS_in of Faster::replan (faster.cpp:370-382): the path up to its first crossing of the sphere of radius min(|goal - start| - 0.001,
    Ra) around its first vertex, the crossing point appended (goal, start: the ends of the path)."""
    path = [np.array(v, dtype=np.float64) for v in path]
    if not Ra > 0 or len(path) < 2:
        return np.array(path)
    ra = min(_norm3(path[-1] - path[0]) - 0.001, Ra)
    for i, v in enumerate(path):
        if _norm3(v - path[0]) > ra:
            if i == 0:
                break
            return np.array(path[:i] + [_sphere_crossing(path[i - 1], path[i], ra, path[0])])
    return np.array(path)


def _shorten_by(path, d):
    acc = 0.0
    for i in range(len(path) - 1, 0, -1):
        v = path[i] - path[i - 1]
        ln = np.sqrt((v * v).sum())
        acc += ln
        if acc > d:
            keep = acc - d
            path = path[:i]
            path.append(path[-1] + v / ln * keep)
            break
    return path


def _norm3(v):
    return np.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])


def safe_path(jps_in, A, R_pos, r_known, drone_radius, max_poly_safe, unknown_pts=None):
    """JPS_safe of Faster::replan: JPS_in cut where it first comes within drone_radius of unknown space (getFirstCollisionJPS against the
    unknown map, :451-452 -> :767-926) and backed off by drone_radius, first vertex replaced by R, at most max_poly_safe legs (:478-490).
    Distance to unknown space: modelled as r_known - |p - A|, or — unknown_pts given (fh_pair_rule mode 2) — the distance to the nearest
    of those points, as the reference's kd-tree returns it (no point at all: the path as it was)."""
    orig = [np.array(v, dtype=np.float64) for v in jps_in]
    cur = [v.copy() for v in orig]
    iteration = 0
    while cur:
        if unknown_pts is not None:
            if len(unknown_pts) == 0:
                break
            d = unknown_pts - cur[0]
            r = float(np.sqrt(np.min(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])))
        else:
            r = max(r_known - _norm3(cur[0] - A), 0.0)
        if r < drone_radius:
            if iteration == 0:
                orig = [orig[0], orig[0] + np.array([0.01, 0.0, 0.0])]
            else:
                eliminated = len(orig) - len(cur) + 1
                orig = orig[:eliminated] + [cur[0]]
                orig = _shorten_by(orig, drone_radius)
            break
        inters, last_id, none_outside = _sphere_exit(cur, r, cur[0])
        if none_outside:
            break
        cur = [inters] + cur[last_id + 1:]
        iteration += 1
    orig[0] = np.array(R_pos, dtype=np.float64)
    return np.array(orig[:max_poly_safe + 1])


def unknown_voxels(origin, res, dims, A, r_known):
    """The mapper's unknown cloud, modelled: centres of the grid cells farther than r_known from A, z-major, x fastest."""
    ix, iy, iz = np.arange(dims[0]), np.arange(dims[1]), np.arange(dims[2])
    x = (ix + 0.5) * res + origin[0]
    y = (iy + 0.5) * res + origin[1]
    z = (iz + 0.5) * res + origin[2]
    Z, Y, X = np.meshgrid(z, y, x, indexing="ij")
    dx, dy, dz = X - A[0], Y - A[1], Z - A[2]
    far = dx * dx + dy * dy + dz * dz > r_known * r_known
    return np.column_stack([X[far], Y[far], Z[far]])
