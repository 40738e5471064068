"""Round 4 study (CPU, fp64, oracle restatement of the controlled scheme): RK4 with the cover pair's conduction integrated
exactly (gl_sc_exp bit 1; bits 2 / 4: lamp exchange, top-compartment air exchange) at smaller nominal sub-step counts, against
the tight fixtures.   python oracle/studies/covexp_study.py [quick]"""
import ctypes, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, '.')
from oracle import gl_oracle as O

COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5,
                   2.6e5, 6e4, 3.2e3, 60.])
def sce(a, b): return np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)
def judge(got, truth, abs_floor=1e-4):
    e = sce(got, truth); bad = e > 1e-4
    floor = bad & (np.abs(got - truth) < abs_floor) & (np.arange(28)[None, :] < 22) & (np.abs(truth) < 1e4 * abs_floor)
    return int((bad & ~floor).any(axis=1).sum()), int(floor.any(axis=1).sum())

G = lambda n: np.load(f'tests/golden/{n}.npz')
p = G('params_default')['p'].astype(np.float64)
L = O.lib()
mask = ctypes.c_int.in_dll(L, 'gl_sc_exp')
pool = ThreadPoolExecutor(8)

def one_step_set(X, U, D, P, n, verify):
    def run(i):
        return O.rk_sc_guarded(X[i], U[i], D[i], P[i] if P is not None else p, 900.0, n, 4, 2, verify=verify)
    return list(pool.map(run, range(len(X))))

def rollout(R, n, nsteps=None):
    acts, w, XR = R['actions'], R['weather'], R['X']
    x = XR[0].copy(); u = np.zeros(6); Xs = [x]; ref = 0; fail = 0
    for k in range(nsteps or len(acts)):
        u = np.clip(u + acts[k].astype(np.float32).astype(np.float64) * np.float64(np.float32(0.1)), 0, 1)
        x, r, ex, f = O.rk_sc_guarded(x, u, w[k], p, 900.0, n, 4, 2)
        Xs.append(x); ref += ex; fail += f
    return O.scaled_rel_err(np.array(Xs), XR[:len(Xs)]), ref, fail

quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
t, st, jp = G('step_tight'), G('step_tight_storm'), G('step_tight_jump')
R10, R3 = G('rollout_10day'), G('rollout_3day_synth')
configs = [(0, 320), (1, 320), (1, 240), (1, 208), (1, 192), (1, 160), (3, 240), (3, 192), (7, 240), (7, 192), (7, 160), (7, 128)]
if quick: configs = [(0, 320), (1, 240), (7, 160)]
for m, n in configs:
    mask.value = m
    t0 = time.time()
    r = one_step_set(t['X'], t['U'], t['D'], t['P'], n, False)
    e_t = sce(np.array([a[0] for a in r]), t['X_tight']).max()
    r = one_step_set(st['X'], st['U'], st['D'], None, n, False)
    gs = np.array([a[0] for a in r]); e_s = sce(gs, st['X_tight']).max(); ws, fs = judge(gs, st['X_tight'])
    sub_s = np.mean([a[2] for a in r]) + n; fail_s = sum(a[3] for a in r)
    r = one_step_set(jp['X'], jp['U'], jp['D'], None, n, True)
    gj = np.array([a[0] for a in r]); wj, fj = judge(gj, jp['X_tight']); fail_j = sum(a[3] for a in r)
    ej = np.quantile(sce(gj, jp['X_tight']).max(axis=1), 0.99)
    r = one_step_set(jp['X'], jp['U'], jp['D'], None, n, False)
    gj = np.array([a[0] for a in r]); wju, fju = judge(gj, jp['X_tight']); fail_ju = sum(a[3] for a in r)
    e3, ref3, f3 = rollout(R3, n)
    e10, ref10, f10 = rollout(R10, n, 300 if quick else None)
    print(f"exp {m} n_sub {n}: tight {e_t:.1e} | storm max {e_s:.1e} >1e-4 {ws} floor {fs} failed {fail_s} mean substeps {sub_s:.0f} | "
          f"jump verified >1e-4 {wj} floor {fj} failed {fail_j} q99 {ej:.1e}; unverified >1e-4 {wju} failed {fail_ju} | "
          f"3-day {e3:.1e} (refined {ref3}, failed {f3}) | 10-day {e10:.1e} (refined {ref10}, failed {f10}) | {time.time()-t0:.0f}s", flush=True)
