"""Round 5 study: where members of the five-stage 2N family fall onto a spurious quasi-steady state of the strongly ventilated top
compartment -- the 43 env-steps the GPU flagged (error estimate) with alpha = 0.0044 at n_sub 120, replayed for alpha = 1/200 ... 0.0042 and n_sub 100 ... 180.
    python oracle/studies/spurious_onset.py  (result: spurious_onset_result.txt)"""
import sys, ctypes
sys.path.insert(0,'.'); sys.path.insert(0,'oracle/studies')
import numpy as np
import lsrk_study as L
from oracle import gl_oracle as O
from concurrent.futures import ThreadPoolExecutor
g=np.load('oracle/studies/r05_flagged_tuples_alpha0044.npz'); X,U,D,F=g['X'],g['U'],g['D'],g['flags']
sel=[i for i in range(len(X)) if (F[i]&0xff)==4]
p=L.p
pool=ThreadPoolExecutor(8)
truth=list(pool.map(lambda i:O.rk4(X[i],U[i],D[i],p,900.0,16384),sel))
ctypes.c_int.in_dll(O.lib(),'gl_ls_exp').value=1
for alpha in (1/200,0.0047,0.0045,0.0044,0.0043,0.0042):
    A,B,S=L.set_scheme(alpha,2)
    out=[]
    for n in (100,108,116,120,124,128,132,140,148,160,180):
        r=list(pool.map(lambda i:O.rk_sc(X[i],U[i],D[i],p,900.0,n,5,2),sel))
        e=np.array([L.sce(r[k][0],truth[k]).max() for k in range(len(sel))]); st=np.array([r[k][1][0] for k in range(len(sel))]); lam=np.array([r[k][1][2] for k in range(len(sel))])
        out.append("n %d (h*lam %.2f, steps %.0f): med %.0e max %.0e"%(n,np.median(lam)*900/n,st.mean(),np.median(e),e.max()))
    print("alpha %.5f S %.3f | "%(alpha,S)+" | ".join(out),flush=True)
