"""Prototype (CPU, fp64, oracle RHS): ETDRK4 with a diagonal linear part for the fast relaxation modes."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O

def phi_coeffs(z, h):
    """Cox-Matthews ETDRK4 coefficients for scalar z = -a*h (vectorised); stable evaluation via series for small |z|."""
    z = np.asarray(z, dtype=np.float64)
    E, E2 = np.exp(z), np.exp(z / 2)
    small = np.abs(z) < 1e-2
    zs = np.where(small, 1.0, z)
    Q = np.where(small, h * (0.5 + z / 8 + z * z / 48), h * (E2 - 1) / zs)
    f1 = np.where(small, h * (1/6 + z/6 + 3*z*z/40), h * (-4 - zs + E * (4 - 3 * zs + zs * zs)) / zs**3)
    f2 = np.where(small, h * (1/3 + z/6 + z*z/20), 2 * h * (2 + zs + E * (-2 + zs)) / zs**3)   # multiplies (Na+Nb)... returned as per-term
    f3 = np.where(small, h * (1/6 - z*z/120), h * (-4 - 3 * zs - zs * zs + E * (4 - zs)) / zs**3)
    return E, E2, Q, f1, f2 / 2, f3     # f2/2 so that x+ = E x + f1 N1 + 2 f2' (Na+Nb)... see step

def rates(x, aux, p, which):
    a = np.zeros(28)
    capCov = 0.1 * np.cos(p[45] * np.pi / 180) * p[73] * p[64] * p[72]
    if 'cover' in which:
        a[6] = 2.0 * (p[71] / p[73]) / capCov          # acts on y6 = x5 - x6
    if 'lamp' in which:
        a[17] = abs(p[185]) / p[184]
    if 'top' in which:
        f = abs(aux[144]) + abs(aux[136])
        r = f / (p[49] - p[48])
        a[3] = r; a[1] = r; a[16] = r
    return a

def to_y(x):
    y = x.copy(); y[5] = x[5] + x[6]; y[6] = x[5] - x[6]; return y
def to_x(y):
    x = y.copy(); x[5] = 0.5 * (y[5] + y[6]); x[6] = 0.5 * (y[5] - y[6]); return x

def etd_step(x0, u, d, p, dt, n_sub, which=('cover', 'lamp', 'top')):
    h = dt / n_sub
    y = to_y(x0)
    def N(yy, a):
        f = O.rhs(to_x(yy), u, d, p)
        fy = f.copy(); fy[5] = f[5] + f[6]; fy[6] = f[5] - f[6]
        return fy + a * yy
    for _ in range(n_sub):
        _, aux = O.rhs(to_x(y), u, d, p, want_aux=True)
        a = rates(to_x(y), aux, p, which)
        E, E2, Q, f1, f2, f3 = phi_coeffs(-a * h, h)
        N1 = N(y, a)
        ya = E2 * y + Q * N1
        Na = N(ya, a)
        yb = E2 * y + Q * Na
        Nb = N(yb, a)
        yc = E2 * ya + Q * (2 * Nb - N1)
        Nc = N(yc, a)
        y = E * y + f1 * N1 + 2 * f2 * (Na + Nb) + f3 * Nc
    return to_x(y)

def sc_err(X, XR):
    sc = np.maximum(np.abs(XR), 1e-3 * np.abs(XR).max(axis=0, keepdims=True)); sc[sc == 0] = 1
    return (np.abs(X - XR) / sc).max()

if __name__ == '__main__':
    p = np.load('tests/golden/params_default.npz')['p'].astype(np.float64)
    R = np.load('tests/golden/rollout_10day.npz'); acts, w, XR = R['actions'], R['weather'], R['X']
    G = np.load('tests/golden/env_rulebased_1day.npz'); Ug, Xg, wg = G['u'], G['x'], G['weather']
    nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    for which in (('cover', 'lamp', 'top'),):
        for n in (32, 48, 64, 96, 128):
            t = time.time()
            x = XR[0].copy(); u = np.zeros(6); Xs = [x]
            bad = False
            for k in range(nsteps):
                u = np.clip(u + acts[k] * np.float32(0.1), 0, 1)
                x = etd_step(x, u, w[k], p, 900.0, n, which); Xs.append(x)
                if not np.all(np.isfinite(x)): bad = True; break
            e1 = np.nan if bad else sc_err(np.array(Xs), XR[:len(Xs)])
            x = Xg[0].copy(); Xs = [x]; bad = False
            for k in range(97):
                x = etd_step(x, Ug[k], wg[k], p, 900.0, n, which); Xs.append(x)
                if not np.all(np.isfinite(x)): bad = True; break
            e2 = np.nan if bad else sc_err(np.array(Xs), Xg[:len(Xs)])
            print(which, 'n_sub', n, 'random-rollout(%d steps) err %.2e | rule-based 1-day err %.2e | %.0fs' % (nsteps, e1, e2, time.time() - t), flush=True)
