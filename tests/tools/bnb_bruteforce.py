"""Oracle branch and bound (with the branching rule of DESIGN.md 4a) against exhaustive enumeration of all P^N assignments, trial by
trial, on 1200 fast safe problems in corridors pulled in by 0.3-0.5 m (6000 trials, feasible and infeasible).  CPU only, ~2 min.
    python tests/tools/bnb_bruteforce.py"""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from faster_amd import abi, corridor
from oracle import oracle
bad=0; ninf=nfe=0
for seed,n_seg,pc,pull,v,a in ((1,6,(3,),0.4,4.8,2.0),(2,5,(2,3),0.5,4.5,1.0),(3,6,(2,3),0.3,4.9,3.0),(4,4,(3,4),0.4,4.0,2.0)):
    pr, faces, verts = corridor.safe_batch(300, seed=700+seed, n_seg=n_seg, p_choices=pc)
    faces=faces.copy(); faces["b"]-=pull
    u = verts[:,1]-verts[:,0]; u/=np.linalg.norm(u,axis=1,keepdims=True)
    pr["x0"][:,3:6]=v*u; pr["x0"][:,6:9]=a*u
    for i in range(len(pr)):
        base=max(oracle.dt_initial(pr[i]), 2*float(pr[i]["dc"]))
        f=1.0
        for _ in range(5):
            dt=f*base
            st,r=oracle.miqp_dt(pr[i],faces,dt)
            nf,bf=oracle.bruteforce_dt(pr[i],faces,dt)
            if nf==0:
                ninf+=1
                if st!=abi.FH_ST_INFEASIBLE: bad+=1; print("BAD inf",seed,i,f,st)
            else:
                nfe+=1
                if st!=abi.FH_ST_OPTIMAL or abs(bf["cost"]-r["cost"])>1e-9*max(1,abs(bf["cost"])): bad+=1; print("BAD",seed,i,f,st,bf["cost"],r["cost"])
            f+=1.0
    print("set",seed,"done: infeasible trials",ninf,"feasible",nfe,"bad",bad,flush=True)
