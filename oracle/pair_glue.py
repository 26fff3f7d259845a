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
