"""One-step stress of the PRODUCT kernels on the GPU: N random (state, control, weather) tuples -- spun-up states, extreme
weather, corner controls, raw control jumps (the distributions of oracle/studies/stress_sc.py, minus its off-trajectory
kind) -- through glgym_evalF.  Truth = the fp64 RK4 kernel with 8x the nominal sub-step count (agrees with 16x to < 2e-7
or the tuple is dropped).  Reports, per scheme and dtype at the DEFAULT sub-step counts, the scaled error distribution.
    python tools/gpu_stress.py [N] [seed]
Evidence for DESIGN.md section 2; not a timed benchmark."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
w = synthetic_weather(n_rows=35040)
COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5,
                   2.6e5, 6e4, 3.2e3, 60.])
sat = lambda t: 610.78 * np.exp(17.2694 * t / (t + 238.3))
D = w[rng.integers(0, 35040, N)].copy()
kind = rng.integers(0, 4, N)                    # 0 plain, 1 extreme weather, 2 corner controls, 3 raw control jump
ex = kind == 1
D[ex, 4] = rng.uniform(0, 40, ex.sum()); D[ex, 1] = rng.uniform(-15, 35, ex.sum()); D[ex, 5] = D[ex, 1] - rng.uniform(0, 25, ex.sum())
D[ex, 2] = rng.uniform(0.3, 1.0, ex.sum()) * sat(D[ex, 1]); D[ex, 0] = np.where(rng.uniform(size=ex.sum()) < 0.5, rng.uniform(0, 1000, ex.sum()), 0.0)
U = rng.uniform(0, 1, (N, 6))
U[kind == 2] = rng.choice([0.0, 1.0], ((kind == 2).sum(), 6))
Uprev = np.clip(U - 0.1 * rng.uniform(-1, 1, (N, 6)), 0, 1)
Uprev[kind == 3] = rng.uniform(0, 1, ((kind == 3).sum(), 6))
X0 = np.array([init_state(d) for d in D])
t0 = time.time()
spin = GreenLight(28, 6, 10, 208, 1800.0, dtype="float64")            # spin-up: 1 800 s under the previous control
XS = spin.evalF_batch(X0, Uprev, D); spin.close()
fine = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", n_sub=2560); T1 = fine.evalF_batch(XS, U, D); fine.close()
finer = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", n_sub=5120); T2 = finer.evalF_batch(XS, U, D); finer.close()
sce = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)
ok = np.isfinite(T2).all(axis=1) & (sce(T1, T2).max(axis=1) < 2e-7)
print(f"{ok.sum()} of {N} tuples with truth ({time.time() - t0:.0f} s); wind up to {D[:, 4].max():.0f} m/s")
for scheme in ("rk4", "rk3", "rk2"):
    for dtype in ("float64", "float32"):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme)
        try:
            Y = m.evalF_batch(XS[ok], U[ok], D[ok]); failed = 0
        except Exception as e:                                    # GLGYM_EODE: rows of failed integrations are NaN
            Y = None; failed = str(e)
        if Y is not None:
            E = sce(Y, T2[ok]).max(axis=1)
            print(f"{scheme} n_sub {m.n_sub} {dtype}: median {np.median(E):.1e}  99 % {np.quantile(E, .99):.1e}  99.9 % "
                  f"{np.quantile(E, .999):.1e}  max {E.max():.1e}  > 1e-4: {(E > 1e-4).sum()}  (by kind {[int(((E > 1e-4) & (kind[ok] == k)).sum()) for k in range(4)]})")
            if "-v" in sys.argv and scheme != "rk2":
                for i in np.argsort(-E)[:3]:
                    j = int(np.argmax(sce(Y[i:i + 1], T2[ok][i:i + 1])))
                    print(f"   worst: err {E[i]:.2e} state {j} (got {Y[i, j]:.6g} truth {T2[ok][i, j]:.6g}) kind {kind[ok][i]} wind {D[ok][i, 4]:.1f} "
                          f"tOut {D[ok][i, 1]:.1f} rad {D[ok][i, 0]:.0f} u {np.round(U[ok][i], 2)} x[2,3,5,6,7,20] {np.round(XS[ok][i, [2, 3, 5, 6, 7, 20]], 2)}")
        else:
            print(scheme, dtype, "failed integrations reported:", failed)
        m.close()
