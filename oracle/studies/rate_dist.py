"""Distribution of the per-env-step maximum of the rate bound on the bench workload (synthetic year, random delta-u-bounded
actions), CPU oracle, with the cover pair's conduction integrated exactly (gl_sc_exp = 1) or not (0).
    python oracle/studies/rate_dist.py [n_envs] [n_steps] [exp] [n_sub] [SC_MOVE]"""
import ctypes, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'greenlight-gym2_amd')
from oracle import gl_oracle as O
from gl_gym_amd.parameters import init_default_params
from gl_gym_amd.utils import synthetic_weather, init_state

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
exp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n_sub = int(sys.argv[4]) if len(sys.argv) > 4 else 240
ctypes.c_int.in_dll(O.lib(), 'gl_sc_exp').value = exp
if len(sys.argv) > 5: ctypes.c_double.in_dll(O.lib(), 'gl_sc_move').value = float(sys.argv[5])
p = init_default_params().astype(np.float64)
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
starts = np.arange(0, 35040 - 5760 - 60, 96)
rng = np.random.default_rng(7)

def run(e):
    r = np.random.default_rng(1000 + e)
    t0 = int(r.choice(starts)) + int(r.integers(0, 5000))
    x = init_state(w[t0]) * (1 + 1e-3 * r.standard_normal(28)); u = np.zeros(6)
    out = []
    for k in range(n_steps):
        a = r.uniform(-1, 1, 6).astype(np.float32)
        u = np.clip(u + a.astype(np.float64) * np.float64(np.float32(0.1)), 0, 1)
        x, st = O.rk_sc(x, u, w[t0 + k], p, 900.0, n_sub, 4, 2)
        out.append((st[2], st[0], st[3], w[t0 + k][4], u[3]))
    return out
t = time.time()
with ThreadPoolExecutor(8) as ex:
    R = np.array([v for o in ex.map(run, range(n_envs)) for v in o])
lam = R[:, 0]
print(f"exp {exp} n_sub {n_sub}: {len(lam)} env-steps in {time.time()-t:.0f}s; rate bound max per env-step: median {np.median(lam):.3f} "
      f"q99 {np.quantile(lam,.99):.3f} q99.9 {np.quantile(lam,.999):.3f} q99.99 {np.quantile(lam,.9999):.3f} max {lam.max():.3f}; "
      f"mean sub-steps {R[:,1].mean():.1f}; flagged {int((R[:,2]!=0).sum())}")
for n in (320, 288, 272, 256, 240, 224, 208, 192):
    thr = 0.92 * 2.785 / (900.0 / n)
    print(f"  n_sub {n}: limit {thr:.3f} 1/s, P(exceeded within an env-step) = {(lam > thr).mean():.2e}")
ns = R[:, 1]
print(f"  sub-steps per env-step: mean {ns.mean():.2f} P(> n_sub) {(ns > n_sub).mean():.3f} q99 {np.quantile(ns,.99):.0f} q99.9 {np.quantile(ns,.999):.0f} q99.99 {np.quantile(ns,.9999):.0f} max {ns.max():.0f} (a launch at one wave per SIMD lasts as long as its slowest lane)")
i = np.argsort(lam)[-5:]
print("  top 5: lam, wind, uVent:", [(round(R[j,0],3), round(R[j,3],1), round(R[j,4],2)) for j in i])
