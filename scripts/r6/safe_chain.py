"""The three stages of a replan that depend on unknown space (fh_pair_rule mode 2), alone and repeated, for a kernel trace:
python scripts/r6/safe_chain.py [pairs] [reps]   (map, paths and whole solves are set up once, untimed)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from faster_amd import abi, capi, corridor, frontend

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.cuda.init()
dev = torch.device("cuda:0")
par = abi.default_params()
N, max_poly, r_known, drone_r, decomp_r, fpp = 6, 3, 4.0, 0.3, 0.05, 96
res, infl, zmax = 0.2, 0.3, 3.0
cloud, cells, center, starts, goals, rng = frontend.forest_queries(pairs, 7, return_rng=True)
B = pairs
whole = abi.make_problems(B)
whole["n_seg"], whole["force_final_pos"], whole["dc"] = N, 1, 0.01
whole["v_max"], whole["a_max"], whole["j_max"] = 5.0, 5.0, 8.0
whole["f_init"], whole["f_final"], whole["f_inc"] = 1.0, 10.0, 1.0
u = goals - starts
u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-9)
whole["x0"][:, 0:3] = starts
whole["x0"][:, 3:6] = u * rng.uniform(0, 1.5, size=(B, 1))
tmpl = corridor.safe_templates(whole)
tmpl["n_seg"] = N


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).to(dev)


ctx, vmap = capi.Context(0), capi.Map(0)
mp = 16
d_cloud, d_starts, d_goals = to_dev(cloud), to_dev(starts), to_dev(goals)
d_whole_t, d_tmpl = to_dev(whole), to_dev(tmpl)
d_whole, d_safe = d_whole_t.clone(), d_tmpl.clone()
f64, i32 = torch.float64, torch.int32
d_paths, d_np, d_ex = torch.zeros((B, mp, 3), dtype=f64, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros(B, dtype=torch.int64, device=dev)
FB, RES = abi.face_dtype.itemsize, abi.result_dtype.itemsize
d_wf = torch.zeros(B * fpp * FB, dtype=torch.uint8, device=dev)
d_off, d_npoly, d_last = torch.zeros((B, 9), dtype=i32, device=dev), torch.zeros(B, dtype=i32, device=dev), torch.zeros((B, 3), dtype=f64, device=dev)
d_wr, d_sr = torch.zeros(B * RES, dtype=torch.uint8, device=dev), torch.zeros(B * RES, dtype=torch.uint8, device=dev)
ctx.set_params(par)
ctx.set_pair_rule(mode=1, r_known=r_known, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
vmap.set_search("jps")
vmap.set_sphere(r_known)
vmap.read_device(d_cloud.data_ptr(), len(cloud), cells, res, center, 0.0, zmax, infl)
vmap.plan_batch_device(d_starts.data_ptr(), d_goals.data_ptr(), B, mp, d_paths.data_ptr(), d_np.data_ptr(), d_ex.data_ptr(), 1.5, 0)
vmap.sync()
dims, origin = vmap.dims()
ctx.corridor_batch_device(d_cloud.data_ptr(), len(cloud), d_paths.data_ptr(), d_np.data_ptr(), B, mp, max_poly, fpp, d_wf.data_ptr(), d_off.data_ptr(),
                          d_npoly.data_ptr(), d_last.data_ptr(), decomp_r, 0.0)
ctx.corridor_problems_device(d_np.data_ptr(), d_last.data_ptr(), d_goals.data_ptr(), d_wf.data_ptr(), d_off.data_ptr(), d_npoly.data_ptr(), B, fpp, N, d_whole.data_ptr())
ctx.solve_batch_device(d_whole.data_ptr(), d_wf.data_ptr(), B, N, fpp, d_wr.data_ptr())
ctx.sync()
iz, iy, ix = np.meshgrid(np.arange(dims[2]), np.arange(dims[1]), np.arange(dims[0]), indexing="ij")
cen = np.stack([(ix + 0.5) * res + origin[0], (iy + 0.5) * res + origin[1], (iz + 0.5) * res + origin[2]], axis=-1)
seen = np.zeros(iz.shape, dtype=bool)
for c in rng.uniform([1, 1, 1.5], [19, 19, 1.5], size=(16, 3)):
    seen |= np.linalg.norm(cen - c, axis=-1) < rng.uniform(2.0, 3.5)
d_flags = to_dev((~seen).astype(np.uint8))
ctx.set_pair_rule(mode=2, drone_radius=drone_r, delta_h=1.0, delta_a=0.5)
ctx.set_unknown_grid_device(d_flags.data_ptr(), origin, res, dims)
fpp2 = 192
d_sf2 = torch.zeros(B * fpp2 * FB, dtype=torch.uint8, device=dev)
t_c, t_s = [], []
for _ in range(reps + 1):
    d_safe.copy_(d_tmpl)
    torch.cuda.synchronize()
    t = time.perf_counter()
    ctx.safe_corridor_batch_device(d_whole.data_ptr(), d_wr.data_ptr(), d_paths.data_ptr(), d_np.data_ptr(), mp, d_goals.data_ptr(), d_cloud.data_ptr(), len(cloud),
                                   origin, res, dims, B, 0.5, max_poly, (2.0, 2.0, 1.0), decomp_r, 0.0, fpp2, N, d_safe.data_ptr(), d_sf2.data_ptr())
    ctx.sync()
    t1 = time.perf_counter()
    ctx.solve_batch_device(d_safe.data_ptr(), d_sf2.data_ptr(), B, N, fpp2, d_sr.data_ptr())
    ctx.sync()
    t2 = time.perf_counter()
    t_c.append(1e3 * (t1 - t)); t_s.append(1e3 * (t2 - t1))
safe = d_safe.cpu().numpy().view(abi.problem_dtype)
sres = d_sr.cpu().numpy().view(abi.result_dtype)
need = safe["n_seg"] > 0
print(json.dumps({"safe_corridor_ms": float(np.median(t_c[1:])), "safe_solve_ms": float(np.median(t_s[1:])), "need": int(need.sum()),
                  "solved_frac": float(sres["solved"][need].mean()), "mean_trials_of_needed": float(sres["trials"][need].mean()),
                  "mean_nodes": float(sres["nodes"][need].mean()), "mean_iters": float(sres["qp_iters"][need].mean()),
                  "mean_faces": float(np.mean([safe["face_off"][i][safe["n_poly"][i]] for i in np.nonzero(need)[0][:4096]])),
                  "mean_npoly": float(safe["n_poly"][need].mean())}))
vmap.close(); ctx.close()
