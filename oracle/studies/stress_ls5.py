#!/usr/bin/env python3
"""Round 5: the wide random stress of stress_sc.py (spun-up states, off-trajectory random states, extreme weather, corner controls, raw
control jumps; truth = plain RK4 at 16 384 ^ 32 768 sub-steps agreeing to 2e-7) for the five-stage 2N scheme against the shipped
exponential RK4, with the movement limiter's allowance scaled by the window's stability head-room to the power q (0 = round 4's limiter).
TEST INFRASTRUCTURE.   python oracle/studies/stress_ls5.py N [seed0] [alpha] [n_sub]"""
import sys, time, ctypes
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "greenlight-gym2_amd")); sys.path.insert(0, str(ROOT / "oracle" / "studies"))
from oracle import gl_oracle as O  # noqa: E402
import stress_sc as SS  # noqa: E402
import lsrk_study as LS  # noqa: E402

alpha = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0047
n_ls = int(sys.argv[4]) if len(sys.argv) > 4 else 128
CFG = [("rk4-240 win4 (round 4)", 240, 4, 4, 0.0), (f"ls5-{n_ls} win2 q0", n_ls, 5, 2, 0.0), (f"ls5-{n_ls} win2 q1", n_ls, 5, 2, 1.0), (f"ls5-{n_ls} win2 q2", n_ls, 5, 2, 2.0),
       ("ls5-192 win1 q1 (parity)", 192, 5, 1, 1.0)]


def tuples(seed):
    """stress_sc.one() without its scheme loop: -> (seed, kind, xs, u, d, truth) or None"""
    rng = np.random.default_rng(seed)
    kind = seed % 5
    w, p, sat, init_state = SS.w, SS.p, SS.sat, SS.init_state
    d = w[int(rng.integers(0, 35040))].copy()
    if kind in (1, 3):
        d[4] = rng.uniform(0, 40); d[1] = rng.uniform(-15, 35); d[5] = d[1] - rng.uniform(0, 25)
        d[2] = rng.uniform(0.3, 1.0) * sat(d[1]); d[0] = rng.uniform(0, 1000) if rng.uniform() < 0.5 else 0.0
    u = rng.uniform(0, 1, 6)
    if kind == 2:
        u = rng.choice([0.0, 1.0], 6)
    u_prev = rng.uniform(0, 1, 6) if kind == 4 else np.clip(u - 0.1 * rng.uniform(-1, 1, 6), 0, 1)
    x0 = init_state(d)
    if kind == 3:
        x0[0:2] = rng.uniform(400, 2500, 2); x0[2:10] += rng.normal(0, 4, 8); x0[17:21] += rng.normal(0, 4, 4)
        x0[9] = rng.uniform(10, 70); x0[15:17] = rng.uniform(0.3, 1.05, 2) * sat(x0[2]); x0[21] = rng.uniform(12, 28)
        x0[22] = rng.uniform(0, 2.5e4); x0[23] = rng.uniform(3e4, 1.1e5); x0[25] = rng.uniform(1e4, 3e5)
        xs = x0
    else:
        xs = O.rk4(x0, u_prev, d, p, float(rng.uniform(300, 5400)), 8192)
    if not np.all(np.isfinite(xs)):
        return None
    a = O.rk4(xs, u, d, p, 900., 16384); b = O.rk4(xs, u, d, p, 900., 32768)
    if not np.all(np.isfinite(b)) or SS.sce(a, b).max() > 2e-7:
        return None
    return seed, kind, xs, u, d, b


if __name__ == "__main__":
    N = int(sys.argv[1]); s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time()
    LS.set_scheme(alpha, 2)
    ctypes.c_int.in_dll(O.lib(), "gl_ls_exp").value = 1
    with ThreadPoolExecutor(8) as ex:
        T = [r for r in ex.map(tuples, range(s0, s0 + N)) if r is not None]
        print(f"{len(T)} tuples with truth of {N}, {time.time() - t0:.0f} s; five-stage scheme: z^5 coefficient {alpha}", flush=True)
        kinds = np.array([t[1] for t in T])
        for name, n, o, wn, q in CFG:
            ctypes.c_double.in_dll(O.lib(), "gl_sc_move_pow").value = q
            R = list(ex.map(lambda t: O.rk_sc_guarded(t[2], t[3], t[4], SS.p, 900., n, o, wn), T))
            E = np.array([SS.sce(r[0], t[5]).max() if np.all(np.isfinite(r[0])) else np.inf for r, t in zip(R, T)])
            failed = np.array([r[3] for r in R]); steps = np.array([r[2] for r in R]) + n; retr = sum(r[1] for r in R)
            bad = (E > 1e-4) & ~failed
            print(f"{name}: err median {np.median(E):.1e} 99% {np.quantile(E, .99):.1e} max {E[~failed].max():.1e}; > 1e-4 (not flagged): {int(bad.sum())} "
                  f"(by kind {[int((bad & (kinds == k)).sum()) for k in range(5)]}); failed {int(failed.sum())}; mean sub-steps {steps.mean():.0f}; retries {retr}", flush=True)
