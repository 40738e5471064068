"""Spectrum of the RHS Jacobian along the fixture rollouts (CPU, oracle RHS): which modes force RK4's 256 sub-steps."""
import sys
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O

def jac(x, u, d, p):
    J = np.zeros((28, 28)); f0 = O.rhs(x, u, d, p)
    for j in range(28):
        h = 1e-6 * max(1.0, abs(x[j])); xp = x.copy(); xp[j] += h; xm = x.copy(); xm[j] -= h
        J[:, j] = (O.rhs(xp, u, d, p) - O.rhs(xm, u, d, p)) / (2 * h)
    return J

p = np.load('tests/golden/params_default.npz')['p'].astype(np.float64)
for fx in ('rollout_10day', 'rollout_3day_synth'):
    g = np.load(f'tests/golden/{fx}.npz'); A, W, X = g['actions'], g['weather'], g['X']
    u = np.zeros(6); worst = 0
    for k in range(0, len(A)):
        u = np.clip(u + A[k] * np.float32(0.1), 0, 1)
        if k % 24: continue
        ev, V = np.linalg.eig(jac(X[k], u, W[k], p))
        o = np.argsort(ev.real)
        lam = ev[o]
        print(fx, k, 'lam:', ' '.join(f'{z.real:.4f}{z.imag:+.3f}j' if abs(z.imag) > 1e-6 else f'{z.real:.4f}' for z in lam[:10]))
        if k % 96 == 0:
            for i in o[:8]:
                v = np.abs(V[:, i]); idx = np.argsort(-v)[:4]
                print('    ', f'{ev[i].real:.4f}', [(int(j), round(float(v[j]), 2)) for j in idx])
