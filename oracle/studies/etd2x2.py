"""Round 4: ETDRK4 on the 2 x 2 cover-pair model -- exponential treatment of the conduction in (sigma, w) coordinates (exact for the
conduction block) against the same on w alone with the coupling into tCovIn left explicit, and against classical RK4: spectral radius
and error of the one-step matrix.   python oracle/studies/etd2x2.py"""
import numpy as np
def phis(z):
    E=np.exp(z); p1=(E-1)/z; p2=(p1-1)/z; p3=(p2-0.5)/z
    return E,p1,p2,p3
def etd_amp(Lmat_diag, J, h):
    # y' = L y + J y (L diag), Cox-Matthews ETDRK4, returns one-step matrix
    n=len(Lmat_diag)
    E=np.ones(n);E2=np.ones(n);Q=np.full(n,h/2);f1=np.full(n,h/6);f2=np.full(n,h/6);f3=np.full(n,h/6)
    for i,a in enumerate(Lmat_diag):
        if a!=0:
            z=a*h
            e,p1,p2,p3=phis(z); E[i]=e
            e2,q1,_,_=phis(z/2); E2[i]=e2; Q[i]=h/2*q1
            f1[i]=h*(p1-3*p2+4*p3); f2[i]=h*(p2-2*p3); f3[i]=h*(4*p3-p2)
    I=np.eye(n)
    N=J
    A=np.diag(E2)+np.diag(Q)@N            # a = A y
    B=np.diag(E2)+np.diag(Q)@N@A          # b
    C=np.diag(E2)@A+np.diag(Q)@(2*N@B-N)
    Y=np.diag(E)+np.diag(f1)@N+2*np.diag(f2)@(N@A+N@B)+np.diag(f3)@N@C
    return Y
g=0.3265
worst=0
for al in [0.0,0.01,0.05,0.1,0.2,0.5]:
  for de in [0.0,0.01,0.05,0.1,0.2]:
    for h in [2.8,3.75,4.5,5.6,7.5]:
      # coords (c,w) diag-only
      J=np.array([[-al,-g],[de-al,-de]])
      Y=etd_amp([0,-2*g],J,h)
      rho=max(abs(np.linalg.eigvals(Y)))
      # exact
      M=np.array([[-al,-g],[de-al,-de-2*g]])
      from scipy.linalg import expm
      Ex=expm(M*h)
      err=np.abs(Y-Ex).max()
      # sigma coords
      J2=np.array([[-(al+de)/2,(-al+de)/2],[(-al+de)/2,-(al+de)/2]])
      Y2=etd_amp([0,-2*g],J2,h)
      M2=J2+np.diag([0,-2*g]); err2=np.abs(Y2-expm(M2*h)).max()
      rho2=max(abs(np.linalg.eigvals(Y2)))
      # classical
      Yc=etd_amp([0,0],M,h); rhoc=max(abs(np.linalg.eigvals(Yc))); errc=np.abs(Yc-Ex).max()
      print(f"al {al:.2f} de {de:.2f} h {h:.2f}: diag(c,w) rho {rho:.3f} err {err:.2e} | (sig,w) rho {rho2:.3f} err {err2:.2e} | classical rho {rhoc:.3f} err {errc:.2e}")
