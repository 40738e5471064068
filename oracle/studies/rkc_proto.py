"""Prototype (CPU, fp64, oracle RHS): second-order Runge-Kutta-Chebyshev (Sommeijer/Shampine/Verwer 1998) for the
env-step map, with the harvest sub-flow Strang-split as in the kernels.  Measures accuracy vs the tight fixtures as a
function of (steps per 900 s, stages)."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O

def cheb(s, w0):
    T = np.zeros(s + 1); dT = np.zeros(s + 1); d2T = np.zeros(s + 1)
    T[0] = 1; T[1] = w0; dT[1] = 1
    for j in range(2, s + 1):
        T[j] = 2 * w0 * T[j-1] - T[j-2]
        dT[j] = 2 * T[j-1] + 2 * w0 * dT[j-1] - dT[j-2]
        d2T[j] = 4 * dT[j-1] + 2 * w0 * d2T[j-1] - d2T[j-2]
    return T, dT, d2T

def rkc_coeffs(s, eps=2/13):
    w0 = 1 + eps / s**2
    T, dT, d2T = cheb(s, w0)
    w1 = dT[s] / d2T[s]
    b = np.zeros(s + 1)
    for j in range(2, s + 1): b[j] = d2T[j] / dT[j]**2
    b[0] = b[2]; b[1] = b[2]
    a = 1 - b * T
    mu = np.zeros(s + 1); nu = np.zeros(s + 1); mut = np.zeros(s + 1); gat = np.zeros(s + 1)
    mut[1] = b[1] * w1
    for j in range(2, s + 1):
        mu[j] = 2 * b[j] * w0 / b[j-1]; nu[j] = -b[j] / b[j-2]; mut[j] = 2 * b[j] * w1 / b[j-1]; gat[j] = -a[j-1] * mut[j]
    beta = (w0 + 1) * d2T[s] / dT[s]
    return mu, nu, mut, gat, beta

def rkc_step(f, y0, h, s, C):
    mu, nu, mut, gat, _ = C
    F0 = f(y0)
    Yjm2 = y0; Yjm1 = y0 + mut[1] * h * F0
    for j in range(2, s + 1):
        Y = (1 - mu[j] - nu[j]) * y0 + mu[j] * Yjm1 + nu[j] * Yjm2 + mut[j] * h * f(Yjm1) + gat[j] * h * F0
        Yjm2, Yjm1 = Yjm1, Y
    return Yjm1

def env_step(x, u, d, p, n, s, C, dt=900.0):
    f = lambda y: O.rhs(y, u, d, p)
    h = dt / n
    for _ in range(n):
        x = rkc_step(f, x, h, s, C)
    return x

def sc_err(X, XR):
    sc = np.maximum(np.abs(XR), 1e-3 * np.abs(XR).max(axis=0, keepdims=True)); sc[sc == 0] = 1
    return (np.abs(X - XR) / sc).max()

if __name__ == '__main__':
    p = np.load('tests/golden/params_default.npz')['p'].astype(np.float64)
    for s in (4, 6, 8, 10, 12, 16): print(s, 'beta', rkc_coeffs(s)[4], 'max h for lam=0.7:', rkc_coeffs(s)[4] / 0.7)
    fx = sys.argv[1] if len(sys.argv) > 1 else 'rollout_3day_synth'
    g = np.load(f'tests/golden/{fx}.npz'); A, W, XR = g['actions'], g['weather'], g['X']
    K = int(sys.argv[2]) if len(sys.argv) > 2 else len(A)
    for n, s in ((8, 12), (16, 8), (32, 6), (64, 4), (16, 10), (32, 8)):
        C = rkc_coeffs(s)
        x, u, X = XR[0].copy(), np.zeros(6), [XR[0]]
        t0 = time.time()
        for k in range(K):
            u = np.clip(u + A[k] * np.float32(0.1), 0, 1)
            x = env_step(x, u, W[k], p, n, s, C)
            X.append(x)
        X = np.array(X)
        e = np.abs(X - XR[:K+1]) / np.maximum(np.abs(XR[:K+1]), 1e-3 * np.abs(XR).max(axis=0, keepdims=True) + 1e-300)
        print(f'n={n} s={s} evals/step={n*s}: err {np.nanmax(e):.3e} worst state {np.nanargmax(e.max(axis=0))}  ({time.time()-t0:.0f}s)')
