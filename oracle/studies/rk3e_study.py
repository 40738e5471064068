"""Round 4 study (CPU, fp64, oracle restatement): the exponential THREE-stage scheme (Cox-Matthews ETD3RK on the cover pair's
conduction mode, Kutta's third-order method on everything else; order 3 of gl_oracle_rk_sc since it became scheme "rk3") against the
exponential RK4 (order 4, n_sub 240) and both with other tier-2b windows, on the tight fixtures.  First run (with Bogacki-Shampine
3(2), conduction in its right-hand side, as order 3 at n_sub 354): rk3e_study_result.txt.
    python oracle/studies/rk3e_study.py"""
import sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O
COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5, 2.6e5, 6e4, 3.2e3, 60.])
def sce(a, b): return np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)
def judge(got, truth, abs_floor=1e-4):
    e = sce(got, truth); bad = e > 1e-4
    floor = bad & (np.abs(got - truth) < abs_floor) & (np.arange(28)[None, :] < 22) & (np.abs(truth) < 1e4 * abs_floor)
    return int((bad & ~floor).any(axis=1).sum()), int(floor.any(axis=1).sum())
G = lambda n: np.load(f'tests/golden/{n}.npz')
p = G('params_default')['p'].astype(np.float64)
pool = ThreadPoolExecutor(8)
t, st, jp = G('step_tight'), G('step_tight_storm'), G('step_tight_jump')
R10, R3 = G('rollout_10day'), G('rollout_3day_synth')
def rollout(R, n, order, win):
    acts, w, XR = R['actions'], R['weather'], R['X']
    x = XR[0].copy(); u = np.zeros(6); Xs = [x]; ref = 0; fail = 0
    for k in range(len(acts)):
        u = np.clip(u + acts[k].astype(np.float32).astype(np.float64) * np.float64(np.float32(0.1)), 0, 1)
        x, r, ex, f = O.rk_sc_guarded(x, u, w[k], p, 900.0, n, order, win)
        Xs.append(x); ref += ex; fail += f
    return O.scaled_rel_err(np.array(Xs), XR), ref, fail
for order, win, n in ((4, 2, 240), (4, 3, 240), (3, 3, 270), (3, 2, 270), (3, 4, 272), (3, 3, 240)):
    t0 = time.time()
    run = lambda X, U, D, P, v: list(pool.map(lambda i: O.rk_sc_guarded(X[i], U[i], D[i], P[i] if P is not None else p, 900.0, n, order, win, verify=v), range(len(X))))
    r = run(t['X'], t['U'], t['D'], t['P'], False); e_t = sce(np.array([a[0] for a in r]), t['X_tight']).max()
    r = run(st['X'], st['U'], st['D'], None, False); gs = np.array([a[0] for a in r]); e_s = sce(gs, st['X_tight']).max(); ws, fs = judge(gs, st['X_tight'])
    sub_s = np.mean([a[2] for a in r]) + n
    r = run(jp['X'], jp['U'], jp['D'], None, True); gj = np.array([a[0] for a in r]); wj, fj = judge(gj, jp['X_tight']); fail_j = sum(a[3] for a in r)
    ej = np.quantile(sce(gj, jp['X_tight']).max(axis=1), 0.99)
    r = run(jp['X'], jp['U'], jp['D'], None, False); gj = np.array([a[0] for a in r]); wju, fju = judge(gj, jp['X_tight'])
    e3, ref3, f3 = rollout(R3, n, order, win); e10, ref10, f10 = rollout(R10, n, order, win)
    stages = (3 if order == 3 else 4) * n
    print(f"order {order} window {win} n_sub {n} ({stages} stages): tight {e_t:.1e} | storm max {e_s:.1e} >1e-4 {ws} mean sub-steps {sub_s:.0f} | jump verified >1e-4 {wj} floor {fj} "
          f"failed {fail_j} q99 {ej:.1e}; unverified >1e-4 {wju} | 3-day {e3:.1e} (refined {ref3}, failed {f3}) | 10-day {e10:.1e} (refined {ref10}, failed {f10}) | {time.time()-t0:.0f}s", flush=True)
