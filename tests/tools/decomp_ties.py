"""Are the polytopes that differ from the reference (tests/tools/decomp_ref_sweep.py, same maps) ties?  Each differing leg is
decomposed again, by both, on three clouds moved by a few float ulps: a tie between equally close obstacle points no longer exists
there, and an implementation difference would remain.  CPU only; needs /root/reference.  PYTHONPATH=. python tests/tools/decomp_ties.py"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from faster_amd import frontend
from oracle.ref_frontend import ref
npaths, nmaps = 40, 120
rng = np.random.default_rng(31)
key = lambda M: M[np.lexsort(np.round(M, 6).T[::-1])]
def same(got, want):
    out=[]
    for (A,b),(A2,b2) in zip(got,want):
        if len(b)!=len(b2): out.append(False); continue
        G,W=key(np.column_stack([A,b])),key(np.column_stack([A2,b2]))
        out.append(len(b)==0 or float(np.abs(G-W).max())<=1e-9)
    return out
diff=fixed=0
prng=np.random.default_rng(5)
for k in range(nmaps):
    side, height = float(rng.choice([10.0, 16.0, 22.0])), float(rng.choice([2.0, 3.0]))
    res, infl = float(rng.choice([0.15, 0.2, 0.25])), float(rng.choice([0.2, 0.3, 0.45]))
    radius = float(rng.choice([0.0, 0.05, 0.2]))
    mvd, max_poly = float(rng.choice([0.8, 1.5, 2.5])), int(rng.choice([3, 5, 8]))
    cloud, _ = frontend.forest_cloud(2000 + k, size=(side, side, height), density=float(rng.choice([0.05, 0.1, 0.2])))
    cloud = cloud.astype(np.float32).astype(np.float64)
    cells, center = (int(side / res) + 6, int(side / res) + 6, int(height / res)), np.array([side / 2, side / 2, height / 2])
    starts = np.column_stack([rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.3, height - 0.3, npaths)])
    goals = np.column_stack([rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.5, side - 0.5, npaths), rng.uniform(0.3, height - 0.3, npaths)])
    frontend.set_search("jps")
    paths, npts, _ = frontend.plan_batch(cloud, cells, res, center, 0.0, height, infl, starts, goals, max_points=max_poly + 1, max_vertex_dist=mvd, max_poly=max_poly)
    frontend.set_search("astar")
    for i in range(npaths):
        if npts[i] < 2: continue
        path = paths[i, :npts[i]]
        want = ref.decompose(path, cloud, radius, 0.0)
        got, _ = frontend.decompose(path, cloud, drone_radius=radius, z_ground=0.0)
        sm = same(got, want)
        for j,ok in enumerate(sm):
            if ok: continue
            diff+=1
            seg = path[j:j+2]
            agree=0
            for rep in range(3):   # the same leg on a cloud moved by a few float ulps: a tie no longer exists
                c2 = (cloud * (1.0 + prng.uniform(-3e-7, 3e-7, cloud.shape))).astype(np.float32).astype(np.float64)
                w2 = ref.decompose(seg, c2, radius, 0.0); g2,_ = frontend.decompose(seg, c2, drone_radius=radius, z_ground=0.0)
                agree += int(all(same(g2,w2)))
            fixed += int(agree==3)
            print("map %d path %d leg %d: rows %d vs %d; equal on 3 perturbed clouds: %d/3"%(k,i,j,len(got[j][1]),len(want[j][1]),agree),flush=True)
print("differing polytopes %d, of which equal to the reference on every perturbed cloud: %d"%(diff,fixed))
