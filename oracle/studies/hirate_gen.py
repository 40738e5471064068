"""Round 5 study: 1 500 one-step tuples of the regime where the five-stage scheme works closest to its limit -- bench-workload states spun up
for two env-steps, then wind 4-26 m/s with the roof vents 0.6-1 open (rate bounds 0.3 ... 1.6 1/s) -- with RK4-16 384 truth -> oracle/studies/_hirate.npz (not committed).
    python oracle/studies/hirate_gen.py [N]"""
import sys, ctypes, time
sys.path.insert(0,'.'); sys.path.insert(0,'oracle/studies'); sys.path.insert(0,'greenlight-gym2_amd')
import numpy as np
import lsrk_study as L
from oracle import gl_oracle as O
from concurrent.futures import ThreadPoolExecutor
from gl_gym_amd.utils import synthetic_weather, init_state
p=L.p; pool=ThreadPoolExecutor(8)
w=synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
rng=np.random.default_rng(5)
N=int(sys.argv[1]) if len(sys.argv)>1 else 1500
def make(i):
    r=np.random.default_rng(100+i)
    t0=int(r.integers(0,30000))
    x=init_state(w[t0])*(1+1e-3*r.standard_normal(28)); u=r.uniform(0,1,6)
    # spin up 2 env-steps under nearby controls
    for k in range(2):
        u=np.clip(u+0.1*r.uniform(-1,1,6),0,1)
        x=O.rk4(x,u,w[t0+k],p,900.0,512)
    d=w[t0+2].copy(); d[4]=r.uniform(4,26)          # wind
    u=np.clip(u+0.1*r.uniform(-1,1,6),0,1); u[3]=r.uniform(0.6,1.0)   # vents wide open
    if r.uniform()<0.5: u[2]=0.0; 
    if r.uniform()<0.5: u[5]=0.0
    truth=O.rk4(x,u,d,p,900.0,16384)
    return x,u,d,truth
t=time.time()
T=list(pool.map(make,range(N)))
X=np.array([a[0] for a in T]);U=np.array([a[1] for a in T]);D=np.array([a[2] for a in T]);XT=np.array([a[3] for a in T])
np.savez('oracle/studies/_hirate.npz',X=X,U=U,D=D,XT=XT)
print("generated",N,"in %.0fs"%(time.time()-t))
