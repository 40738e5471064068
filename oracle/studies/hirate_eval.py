"""Round 5 study: members of the five-stage 2N family (z^5 coefficient alpha, usable interval S, n_sub) and the shipped RK4-240 on the
high-rate tuples of hirate_gen.py plus the tuples the GPU flagged with alpha = 0.0044 (r05_flagged_tuples_alpha0044.npz).
    python oracle/studies/hirate_eval.py"""
import sys, ctypes, time
sys.path.insert(0,'.'); sys.path.insert(0,'oracle/studies')
import numpy as np
import lsrk_study as L
from oracle import gl_oracle as O
from concurrent.futures import ThreadPoolExecutor
p=L.p; pool=ThreadPoolExecutor(8)
g=np.load('oracle/studies/_hirate.npz'); X,U,D,XT=g['X'],g['U'],g['D'],g['XT']
f=np.load('oracle/studies/r05_flagged_tuples_alpha0044.npz'); sel=[i for i in range(len(f['X'])) if (f['flags'][i]&0xff)==4]
XF,UF,DF=f['X'][sel],f['U'][sel],f['D'][sel]
XTF=np.array(list(pool.map(lambda i:O.rk4(XF[i],UF[i],DF[i],p,900.0,16384),range(len(XF)))))
X=np.concatenate([X,XF]);U=np.concatenate([U,UF]);D=np.concatenate([D,DF]);XT=np.concatenate([XT,XTF])
ctypes.c_int.in_dll(O.lib(),'gl_ls_exp').value=1
dp=ctypes.POINTER(ctypes.c_double)
def run(order,n,win):
    r=list(pool.map(lambda i:O.rk_sc(X[i],U[i],D[i],p,900.0,n,order,win),range(len(X))))
    e=np.array([L.sce(r[k][0],XT[k]).max() for k in range(len(X))]); st=np.array([r[k][1][0] for k in range(len(X))]); lam=np.array([r[k][1][2] for k in range(len(X))]); fl=np.array([int(r[k][1][3]) for k in range(len(X))]); est=np.array([r[k][1][1] for k in range(len(X))])
    return e,st,lam,fl,est
e,st,lam,fl,est=run(4,240,4)
print(f"rk4 240/4: lam median {np.median(lam):.2f} q90 {np.quantile(lam,.9):.2f} max {lam.max():.2f}; err med {np.median(e):.1e} q99 {np.quantile(e,.99):.1e} max {e.max():.1e}; >3e-5: {(e>3e-5).sum()} >1e-4: {(e>1e-4).sum()}; steps mean {st.mean():.0f}; flagged {(fl!=0).sum()}")
for alpha,Ss in ((0.0044,(5.4588,4.8,4.4,4.2)),(0.0047,(5.009,4.8,4.6)),(0.005,(4.657,))):
    A,B,S0=L.set_scheme(alpha,2)
    for S in Ss:
        O.lib().gl_oracle_set_lsrk(np.ascontiguousarray(A).ctypes.data_as(dp),np.ascontiguousarray(B).ctypes.data_as(dp),float(S),2)
        for n in (120,128,136,144):
            e,st,lam,fl,est=run(5,n,2)
            bad=(e>3e-5); unfl_bad=bad&(fl==0)
            print(f"alpha {alpha} S {S} n {n}: err med {np.median(e):.1e} q99 {np.quantile(e,.99):.1e} max {e.max():.1e}; >3e-5: {bad.sum()} (unflagged {unfl_bad.sum()}, worst unflagged {e[fl==0].max():.1e}) >1e-4: {(e>1e-4).sum()}; steps mean {st.mean():.0f} max {st.max():.0f}; flagged {(fl!=0).sum()}",flush=True)
