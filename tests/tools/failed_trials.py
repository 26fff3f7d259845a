"""CPU study (oracle = test infrastructure; this script is an experiment, not product): why do the failed trials of C4's safe problems fail?
For every failed trial of a sample of safe problems: refuted at y = 0 by the jerk box (the kernel's early exit)?  infeasible WITHOUT the
corridor (n_poly = 0: the v/a/j boxes and the final-state equalities alone)?  refuted by the per-axis stopping-time bound?"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from faster_amd import abi, corridor
from oracle import oracle, pair_glue

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
whole, faces, _ = corridor.whole_batch(B, seed=3, n_seg=10, p_choices=(2, 3, 4, 5, 6))
wres = oracle.solve_batch(whole, faces)
safe_t = corridor.safe_templates(whole)
safe, sfaces = pair_glue.glue(whole, wres, faces, safe_t, 0.5, 0.2, 3, r_margin=0.05)
sres = oracle.solve_batch(safe, sfaces)
print("whole solved %.3f safe solved %.3f trials %.2f / %.2f" % (wres["solved"].mean(), sres["solved"].mean(), wres["trials"].mean(), sres["trials"].mean()))


def stop_time(v0, a0, J, A):
    """minimal time to bring (v, a) to (0, 0) with |jerk| <= J and |a| <= A (continuous time, bang-(coast)-bang)"""
    # mirror so that the required velocity change dv = -v_eff is handled with sign s
    best = np.inf
    # velocity after bringing a0 to zero at full jerk: v1 = v0 + a0|a0|/(2J)
    v1 = v0 + a0 * abs(a0) / (2 * J)
    s = -np.sign(v1) if v1 != 0 else 0.0
    if s == 0:
        return abs(a0) / J
    # accelerate with jerk s J to a peak ap (sign s), optionally hold at s A, return with jerk -s J to 0
    # velocity change: from a0 to ap: (ap^2 - a0^2)/(2 s J); from ap to 0: ap^2/(2 s J) -> total (2 ap^2 - a0^2)/(2 s J) = -v0
    ap2 = (a0 * a0 - 2 * s * J * v0) / 2.0  # ap^2 with dv = -v0:  (2 ap^2 - a0^2) / (2 s J) = -v0  => ap^2 = (a0^2 - 2 s J v0) / 2
    ap = np.sqrt(max(ap2, 0.0))
    if ap <= A:
        return (abs(s * ap - a0) + ap) / J
    # saturated: ramp a0 -> s A, hold th, ramp s A -> 0
    dv_ramps = (A * A - a0 * a0) / (2 * s * J) + A * A / (2 * s * J)
    th = (-v0 - dv_ramps) / (s * A)
    return (abs(s * A - a0) + A) / J + max(th, 0.0)


par = abi.default_params()
n_failed = n_y0 = n_box = n_stop = n_stop_only = 0
per_problem = []
for i in range(B):
    ps = safe[i]
    if ps["n_seg"] == 0:
        continue
    base = max(oracle.dt_initial(ps), 2 * ps["dc"])
    nfail = int(sres[i]["trials"]) - (1 if sres[i]["solved"] else 0)
    f = float(ps["f_init"])
    nobox = ps.copy()
    nobox["n_poly"] = 0
    N = int(ps["n_seg"])
    for k in range(nfail):
        dt = f * base
        n_failed += 1
        st, r = oracle.miqp_dt(nobox, sfaces, dt)
        box_inf = r["solved"] == 0
        n_box += box_inf
        T = max(stop_time(ps["x0"][3 + a], ps["x0"][6 + a], ps["j_max"] + 1e-9, ps["a_max"] + 1e-9) for a in range(3))
        stop_inf = N * dt < T * (1 - 1e-9)
        n_stop += stop_inf
        if stop_inf and not box_inf:
            n_stop_only += 1
        f = f + float(ps["f_inc"])
print("failed safe trials %d (%.2f per pair): infeasible without the corridor %d (%.1f %%), refuted by the stopping-time bound %d (%.1f %%), bound fired on a box-feasible trial %d"
      % (n_failed, n_failed / B, n_box, 100.0 * n_box / max(n_failed, 1), n_stop, 100.0 * n_stop / max(n_failed, 1), n_stop_only))
