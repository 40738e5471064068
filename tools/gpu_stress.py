"""One-step stress of the PRODUCT kernels on the GPU: N random (state, control, weather) tuples -- spun-up states, extreme
weather, corner controls, raw control jumps (the distributions of oracle/studies/stress_sc.py, minus its off-trajectory
kind), and -- round 3 -- the round-2 review's recipe (oracle/studies/stress_jump.py: vents slammed to 1, screens pulled to 0,
cold, 8-40 m/s wind, the row's vapour pressure) -- through glgym_evalF, which integrates step-doubling VERIFIED by default
(`python tools/gpu_stress.py N seed never` runs the unverified guard instead).  Truth = the fp64 RK4 kernel with 8x the nominal sub-step count (agrees with 16x to < 2e-7
or the tuple is dropped).  Reports, per scheme and dtype at the DEFAULT sub-step counts, the scaled error distribution.
    python tools/gpu_stress.py [N] [seed]
Evidence for DESIGN.md section 2; not a timed benchmark."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state


def run_stress(N=32768, seed=7, schemes=("ls5", "rk4", "rk3", "rk2"), dtypes=("float64", "float32"), argv=()):
    """-> {(scheme, dtype): {"n": tuples with truth, "n_kind": per kind, "real": per kind (> 1e-4 beyond the metric floor), "gross": per kind
    (> 1e-2), "failed": per kind, "q999": 99.9 % quantile}}; prints the tool's report lines."""
    sys_argv = list(argv)
    rng = np.random.default_rng(seed)
    results = {}
    w = synthetic_weather(n_rows=35040)
    COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5,
                       2.6e5, 6e4, 3.2e3, 60.])
    sat = lambda t: 610.78 * np.exp(17.2694 * t / (t + 238.3))
    D = w[rng.integers(0, 35040, N)].copy()
    kind = rng.integers(0, 5, N)                    # 0 plain, 1 extreme weather, 2 corner controls, 3 raw control jump, 4 the review's recipe
    ex = kind == 1
    D[ex, 4] = rng.uniform(0, 40, ex.sum()); D[ex, 1] = rng.uniform(-15, 35, ex.sum()); D[ex, 5] = D[ex, 1] - rng.uniform(0, 25, ex.sum())
    D[ex, 2] = rng.uniform(0.3, 1.0, ex.sum()) * sat(D[ex, 1]); D[ex, 0] = np.where(rng.uniform(size=ex.sum()) < 0.5, rng.uniform(0, 1000, ex.sum()), 0.0)
    U = rng.uniform(0, 1, (N, 6))
    U[kind == 2] = rng.choice([0.0, 1.0], ((kind == 2).sum(), 6))
    Uprev = np.clip(U - 0.1 * rng.uniform(-1, 1, (N, 6)), 0, 1)
    Uprev[kind == 3] = rng.uniform(0, 1, ((kind == 3).sum(), 6))
    jr = kind == 4
    nj = int(jr.sum())
    D[jr, 1] = rng.uniform(-8, 8, nj); D[jr, 5] = D[jr, 1] - rng.uniform(5, 20, nj)
    D[jr, 0] = np.where(rng.uniform(size=nj) < 0.6, 0.0, rng.uniform(0, 300, nj)); D[jr, 4] = rng.uniform(8, 40, nj)
    Uprev[jr] = np.column_stack([rng.uniform(.3, 1, nj), rng.uniform(size=nj), rng.uniform(size=nj), rng.uniform(0, .2, nj), rng.uniform(size=nj), rng.uniform(size=nj)])
    U[jr] = np.column_stack([np.where(rng.uniform(size=nj) < .5, 0.0, Uprev[jr, 0]), rng.choice([0.0, 1.0], nj), np.zeros(nj), np.ones(nj), rng.choice([0.0, 1.0], nj), np.zeros(nj)])
    X0 = np.array([init_state(d) for d in D])
    t0 = time.time()
    spin = GreenLight(28, 6, 10, 208, 1800.0, dtype="float64"); spin.set_verify("never")          # spin-up: 1 800 s under the previous control
    import ctypes as C
    from gl_gym_amd import _lib as L
    def raw_evalF(m, X, U_, D_):
        """glgym_evalF without the exception: rows of failed integrations come back NaN, the others valid"""
        xs_, u_, d_ = (np.ascontiguousarray(a, dtype=np.float64) for a in (X, U_, D_))
        Y = np.empty_like(xs_)
        rc = m._lib.glgym_evalF(m._h, xs_.ctypes.data_as(L._DP), u_.ctypes.data_as(L._DP), d_.ctypes.data_as(L._DP), None, 1, len(xs_), Y.ctypes.data_as(L._DP))
        assert rc in (0, L.EODE), rc
        return Y
    XS = raw_evalF(spin, X0, Uprev, D); spin.close()
    spun = ~np.isnan(XS).any(axis=1)
    print(f"spin-up (1 800 s from the reset state, unverified guard): {int((~spun).sum())} of {N} rows reported as failed integrations (by kind {[int((~spun & (kind == k)).sum()) for k in range(5)]}); dropped")
    XS, U, D, kind = XS[spun], U[spun], D[spun], kind[spun]; N = len(XS)
    fine = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme="rk4", n_sub=2560); fine.set_verify("never"); T1 = raw_evalF(fine, XS, U, D); fine.close()
    finer = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme="rk4", n_sub=5120); finer.set_verify("never"); T2 = raw_evalF(finer, XS, U, D); finer.close()
    sce = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)
    ok = np.isfinite(T2).all(axis=1) & np.isfinite(T1).all(axis=1) & (sce(np.nan_to_num(T1), np.nan_to_num(T2)).max(axis=1) < 2e-7)
    print(f"truth runs (fp64 kernel, n_sub 2 560 / 5 120, unverified guard): {int(np.isnan(T1).any(axis=1).sum())} / {int(np.isnan(T2).any(axis=1).sum())} rows reported as failed integrations")
    print(f"{ok.sum()} of {N} tuples with truth ({time.time() - t0:.0f} s); wind up to {D[:, 4].max():.0f} m/s")
    if "dump4" in sys_argv:        # the review-recipe tuples with their truth, for offline studies of the guard (oracle restatement)
        k4 = ok & (kind == 4)
        np.savez("gpurun_out/r03_kind4_tuples.npz", x=XS[k4], u=U[k4], d=D[k4], truth=T2[k4])
        print(f"dumped {int(k4.sum())} review-recipe tuples"); return results
    for scheme in schemes:
        for dtype in dtypes:
            m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, preset="throughput")     # the schemes' nominal counts
            if "never" in sys_argv:
                m.set_verify("never")
            Y = raw_evalF(m, XS[ok], U[ok], D[ok]); rc = 0
            failed = int(np.isnan(Y).any(axis=1).sum())
            good = ~np.isnan(Y).any(axis=1)
            if rc in (0, L.EODE):
                E = np.where(good, sce(Y, T2[ok]).max(axis=1), 0.0)
                A = np.where(good, np.abs(Y - T2[ok])[np.arange(len(Y)), sce(np.nan_to_num(Y), T2[ok]).argmax(axis=1)], 0.0)
                real = (E > 1e-4) & ~((A < (1e-4 if dtype == "float64" else 2e-4)) & (sce(np.nan_to_num(Y), T2[ok]).argmax(axis=1) < 22))
                print(f"{scheme} n_sub {m.n_sub} {dtype}: median {np.median(E):.1e}  99 % {np.quantile(E, .99):.1e}  99.9 % "
                      f"{np.quantile(E, .999):.1e}  max {E.max():.1e}  > 1e-4: {(E > 1e-4).sum()} (real, beyond the metric floor: {int(real.sum())}; gross > 1e-2: "
                      f"{int((E > 1e-2).sum())}; by kind {[int((real & (kind[ok] == k)).sum()) for k in range(5)]})  failed {failed} (by kind {[int((~good & (kind[ok] == k)).sum()) for k in range(5)]})")
                kk = kind[ok]
                results[(scheme, dtype)] = {"n": int(ok.sum()), "n_kind": [int((kk == k).sum()) for k in range(5)],
                                            "real": [int((real & (kk == k)).sum()) for k in range(5)],
                                            "gross": [int(((E > 1e-2) & (kk == k)).sum()) for k in range(5)],
                                            "failed": [int((~good & (kk == k)).sum()) for k in range(5)], "q999": float(np.quantile(E, .999))}
                if "keep" in sys_argv and scheme == "rk4" and dtype == "float64" and ((E > 1e-4).any() or failed):      # keep the offenders for an offline look
                    keep = (E > 1e-4) | ~good
                    np.savez("gpurun_out/r03_gpu_stress_offenders.npz", x=XS[ok][keep], u=U[ok][keep], d=D[ok][keep], truth=T2[ok][keep],
                             got=Y[keep], kind=kind[ok][keep])
                if "-v" in sys_argv and scheme != "rk2":
                    for i in np.argsort(-E)[:3]:
                        j = int(np.argmax(sce(Y[i:i + 1], T2[ok][i:i + 1])))
                        print(f"   worst: err {E[i]:.2e} state {j} (got {Y[i, j]:.6g} truth {T2[ok][i, j]:.6g}) kind {kind[ok][i]} wind {D[ok][i, 4]:.1f} "
                              f"tOut {D[ok][i, 1]:.1f} rad {D[ok][i, 0]:.0f} u {np.round(U[ok][i], 2)} x[2,3,5,6,7,20] {np.round(XS[ok][i, [2, 3, 5, 6, 7, 20]], 2)}")
            else:
                print(scheme, dtype, "glgym_evalF status", rc)
            m.close()

    return results


if __name__ == "__main__":
    run_stress(int(sys.argv[1]) if len(sys.argv) > 1 else 32768, int(sys.argv[2]) if len(sys.argv) > 2 else 7, argv=sys.argv[3:])
