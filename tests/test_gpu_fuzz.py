"""Random parity fuzz: glgym_evalF (fp64 and fp32, both sub-steppers) against the oracle's restatement of the same scheme on
random (state, control, weather, parameter) tuples far off the fixture trajectories -- every if_else branch, both harvest
regimes, calm and storm (some tuples are refined by the stability control), noisy crop parameter blocks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tuples(n, golden):
    from gl_gym_amd.utils import synthetic_weather, init_state
    rng = np.random.default_rng(20261003)
    p0 = golden("params_default")["p"].astype(np.float64)
    w = synthetic_weather(35040)
    X, U, D, P = [], [], [], []
    for i in range(n):
        d = w[rng.integers(0, len(w))].copy()
        if i % 9 == 0:
            d[4] = rng.uniform(0.0, 0.3)             # leakage branch (wind below threshold)
        if i % 11 == 0:
            d[4] = rng.uniform(12, 17)               # strong wind
        x = init_state(d)
        x[0:2] = rng.uniform(500, 2500, 2); x[2:10] += rng.normal(0, 3, 8); x[17:21] += rng.normal(0, 3, 4)
        x[9] = rng.uniform(15, 70); x[10:15] += rng.normal(0, 2, 5)
        x[15:17] *= rng.uniform(0.5, 1.1, 2); x[21] = rng.uniform(14, 26)
        x[22] = rng.uniform(-5, 2.5e4); x[23] = rng.uniform(3e4, 1.2e5); x[24] = rng.uniform(1e5, 4e5)
        x[25] = rng.uniform(1e4, 3.3e5); x[26] = rng.uniform(-500, 4000)
        u = rng.choice([0.0, 1.0], 6) if i % 4 == 0 else rng.uniform(0, 1, 6)
        p = p0.copy()
        if i % 3 == 0:                                # noise.py-style crop block, float32 arithmetic
            f = (1 + rng.uniform(-0.1, 0.1, 34)).astype(np.float32)
            p[128:162] = (p[128:162].astype(np.float32) * f).astype(np.float64)
            p[144] = np.float64(np.float32(p[141]) / np.float32(p[142]))
        X.append(x); U.append(u); D.append(d); P.append(p)
    return map(np.array, (X, U, D, P))


@pytest.mark.parametrize("scheme,order,win64,win32,n_sub", [("ls5", 5, 2, 2, 128), ("rk4", 4, 4, 4, 256), ("rk2", 2, 4, 4, 360), ("rk3", 3, 3, 3, 282)])
def test_random_tuples_against_oracle_scheme(golden, oracle, scheme, order, win64, win32, n_sub):
    from gl_gym_amd import GreenLight
    N = 400
    X, U, D, P = _tuples(N, golden)
    scale = np.maximum(np.abs(X).max(axis=0), 1e-3)
    for dtype, win, tol in (("float64", win64, 1e-11), ("float32", win32, 5e-6)):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, n_sub=n_sub)
        worst, guarded = 0.0, 0
        for i in range(N):
            ref, retries, refined, failed = oracle.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, n_sub, order, win, verify=True)   # evalF integrates verified
            assert not failed
            guarded += (refined > 0) or (retries > 0)  # tuples on which the stability control / the guard acted
            got = np.array(m.evalF(X[i], U[i], D[i], P[i]))
            assert np.all(np.isfinite(got)), (scheme, dtype, i)
            worst = max(worst, float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), scale))))
        print(f"fuzz {scheme} {dtype}: worst scaled |product - oracle| over {N} tuples = {worst:.2e}, {guarded} refined / retried")
        assert worst < tol, (scheme, dtype, worst)
        assert guarded >= 1                           # the sample does exercise the stability control
        m.close()
