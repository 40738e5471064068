"""Prototype (CPU, fp64, oracle RHS): Rosenbrock-W (ROS34PW2, Rang & Angermann 2005: order 3 for ANY matrix J, L-stable,
stiffly accurate) with J = the Jacobian block of a small stiff subset S, frozen over the env-step.
Questions: which S, how many sub-steps m, what accuracy vs the tight fixtures."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O

G = 4.3586652150845900e-01
AL = np.zeros((4, 4)); GA = np.zeros((4, 4))
AL[1, 0] = 8.7173304301691801e-01
AL[2, 0] = 8.4457060015369423e-01; AL[2, 1] = -1.1299064236484185e-01
AL[3, 2] = 1.0
GA[1, 0] = -8.7173304301691801e-01
GA[2, 0] = -9.0338057013044082e-01; GA[2, 1] = 5.4180672388095326e-02
GA[3, 0] = 2.4212380706095346e-01; GA[3, 1] = -1.2232505839045147e+00; GA[3, 2] = 5.4526025533510214e-01
BB = np.array([2.4212380706095346e-01, -1.2232505839045147e+00, 1.5452602553351020e+00, 4.3586652150845900e-01])

def rosw_step(f, y, h, J, Minv):
    k = []
    for i in range(4):
        yi = y + h * sum(AL[i, j] * k[j] for j in range(i)) if i else y
        r = f(yi)
        if i:
            r = r + h * (J @ sum(GA[i, j] * k[j] for j in range(i)))
        k.append(Minv @ r)
    return y + h * sum(BB[i] * k[i] for i in range(4))

def order_check():
    rng = np.random.default_rng(0)
    A = rng.normal(size=(4, 4)) - 2 * np.eye(4)
    f = lambda y: A @ y + np.sin(y) * np.array([1, .5, -1, 2]) + 0.3 * y[::-1] ** 2
    from scipy.integrate import solve_ivp
    y0 = np.array([0.3, -0.2, 0.5, 0.1]); T = 1.0
    ref = solve_ivp(lambda t, y: f(y), (0, T), y0, rtol=1e-13, atol=1e-14, method='DOP853').y[:, -1]
    for name, J in (('J=0', np.zeros((4, 4))), ('J=rand', rng.normal(size=(4, 4))), ('J=A', A)):
        errs = []
        for n in (20, 40, 80):
            h = T / n; Minv = np.linalg.inv(np.eye(4) - h * G * J); y = y0.copy()
            for _ in range(n): y = rosw_step(f, y, h, J, Minv)
            errs.append(np.abs(y - ref).max())
        print(name, errs, 'orders', np.log2(errs[0] / errs[1]), np.log2(errs[1] / errs[2]))

def fd_jac(f, x, S):
    f0 = f(x); J = np.zeros((28, 28))
    for j in S:
        h = 1e-5 * max(1.0, abs(x[j])); xp = x.copy(); xp[j] += h
        J[:, j] = (f(xp) - f0) / h
    return J

def env_step(x, u, d, p, m, S, dt=900.0, refresh=0):
    f = lambda y: O.rhs(y, u, d, p)
    h = dt / m
    mask = np.zeros((28, 28)); mask[np.ix_(S, S)] = 1
    for i in range(m):
        if i == 0 or (refresh and i % refresh == 0):
            J = fd_jac(f, x, S) * mask
            Minv = np.linalg.inv(np.eye(28) - h * G * J)
        x = rosw_step(f, x, h, J, Minv)
    return x

if __name__ == '__main__':
    order_check()
    p = np.load('tests/golden/params_default.npz')['p'].astype(np.float64)
    fx = sys.argv[1] if len(sys.argv) > 1 else 'rollout_3day_synth'
    g = np.load(f'tests/golden/{fx}.npz'); A, W, XR = g['actions'], g['weather'], g['X']
    K = int(sys.argv[2]) if len(sys.argv) > 2 else len(A)
    ALL = list(range(28))
    S9 = [1, 3, 5, 6, 7, 16, 17, 20, 15]
    S12 = S9 + [0, 2, 4]
    # refresh = 1: finite-difference Jacobian re-evaluated at every sub-step; 0: frozen at the start of the env-step
    for name, S, ms, refresh in (('full, J refreshed per sub-step', ALL, (8, 16, 32, 64), 1),
                                 ('9 fast states, J refreshed', S9, (16, 32, 64), 1),
                                 ('full, J frozen over the env-step', ALL, (32, 64), 0),
                                 ('12 states, J frozen', S12, (32, 64), 0)):
        for m in ms:
            x, u, X = XR[0].copy(), np.zeros(6), [XR[0]]
            t0 = time.time()
            with np.errstate(all='ignore'):
                for k in range(K):
                    u = np.clip(u + A[k] * np.float32(0.1), 0, 1)
                    x = env_step(x, u, W[k], p, m, S, refresh=refresh)
                    X.append(x)
            X = np.array(X)
            e = np.abs(X - XR[:K+1]) / np.maximum(np.abs(XR[:K+1]), 1e-3 * np.abs(XR).max(axis=0, keepdims=True) + 1e-300)
            print(f'{name}, m={m}: err {np.max(e):.3e} (nan = diverged)  ({time.time()-t0:.0f}s)', flush=True)
