"""One-off fuzz: glgym_evalF (fp64 and fp32, both schemes) against the oracle's restatement of the same scheme on
random (state, control, weather, parameter) tuples far off the fixture trajectories -- every if_else branch, both harvest
regimes, calm and storm, noisy crop parameters.  Prints the worst scaled deviation per configuration."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state
from oracle import gl_oracle as O

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(20261003)
p0 = np.load("tests/golden/params_default.npz")["p"].astype(np.float64)
w = synthetic_weather(35040)
X, U, D, P = [], [], [], []
for i in range(N):
    d = w[rng.integers(0, len(w))].copy()
    if i % 9 == 0: d[4] = rng.uniform(0.0, 0.3)            # leakage branch (wind below threshold)
    if i % 11 == 0: d[4] = rng.uniform(12, 17)             # strong wind
    x = init_state(d)
    x[0:2] = rng.uniform(500, 2500, 2); x[2:10] += rng.normal(0, 3, 8); x[17:21] += rng.normal(0, 3, 4)
    x[9] = rng.uniform(15, 70); x[10:15] += rng.normal(0, 2, 5)
    x[15:17] *= rng.uniform(0.5, 1.1, 2); x[21] = rng.uniform(14, 26)
    x[22] = rng.uniform(-5, 2.5e4); x[23] = rng.uniform(3e4, 1.2e5); x[24] = rng.uniform(1e5, 4e5)
    x[25] = rng.uniform(1e4, 3.3e5); x[26] = rng.uniform(-500, 4000)
    u = rng.choice([0.0, 1.0], 6) if i % 4 == 0 else rng.uniform(0, 1, 6)
    p = p0.copy()
    if i % 3 == 0:
        f = (1 + rng.uniform(-0.1, 0.1, 34)).astype(np.float32)
        p[128:162] = (p[128:162].astype(np.float32) * f).astype(np.float64); p[144] = np.float64(np.float32(p[141]) / np.float32(p[142]))
    X.append(x); U.append(u); D.append(d); P.append(p)
X, U, D, P = map(np.array, (X, U, D, P))
scale = np.maximum(np.abs(X).max(axis=0), 1e-3)
for scheme, order, win64, win32, n in (("rk4", 4, 1, 2, 256), ("rk2", 2, 4, 4, 360)):
    for dtype, win, tol in (("float64", win64, 1e-9), ("float32", win32, 1e-4)):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme)
        worst, where, nonfinite = 0.0, -1, 0
        for i in range(N):
            ref = O.rk_lagged(X[i], U[i], D[i], P[i], 900.0, n, order, win)
            got = np.array(m.evalF(X[i], U[i], D[i], P[i]))
            if not np.all(np.isfinite(ref)):
                nonfinite += 1                      # plain scheme overflowed: the kernel's guard redoes the step with
                for mult in (2, 4):                 # 2x, then 4x sub-steps -- so does this reference
                    ref = O.rk_lagged(X[i], U[i], D[i], P[i], 900.0, n * mult, order, win)
                    if np.all(np.isfinite(ref)):
                        break
            e = np.max(np.abs(got - ref) / np.maximum(np.abs(ref), scale))
            if not np.isfinite(e) or e > worst: worst, where = e, i
        print(f"{scheme} {dtype}: worst scaled |product - oracle| over {N} tuples = {worst:.2e} (tuple {where}); "
              f"{nonfinite} tuples needed the stability guard (compared at 2x / 4x sub-steps); {'OK' if worst < tol else 'CHECK'}", flush=True)
        m.close()
