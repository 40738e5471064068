"""Round 4: how long may the tier-2b window of the RK4 scheme be?  fp64 restatement of the controlled scheme (oracle/gl_oracle.c) at
n_sub 240 with windows of 3 ... 6 sub-steps (11.25 ... 22.5 s) against the tight fixtures: the 10-day rollout, the one-step tuples
(with the tuple that carries the maximum singled out), the storm and raw-jump fixtures in verified mode; and the parity
configuration's n_sub at window 4.  CPU only, ~2 min.      python oracle/studies/window_study.py
Result: window 4 keeps every fixture <= 6.2e-5; window 5 puts the empty-buffer tuple at 9.6e-5."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import gl_oracle as O
O.build()
g = np.load("tests/golden/rollout_10day.npz")
A, W, XR = g["actions"], g["weather"], g["X"]
p = np.load("tests/golden/params_default.npz")["p"].astype(np.float64)
scale = np.maximum(np.abs(XR), 1e-3 * np.abs(XR).max(axis=0, keepdims=True)); scale[scale == 0] = 1
for win in (3, 4, 5, 6):
    x = XR[0].copy(); u = np.zeros(6); worst = 0; nst = 0; fl = 0
    for k in range(len(A)):
        u = np.clip(u + A[k] * np.float32(0.1), 0, 1)
        r = O.rk_sc_guarded(x, u, W[k], p, 900.0, 240, 4, win, want_flags=True)
        x = r[0]; nst += r[2]; fl += int(r[1] > 0)
        worst = max(worst, np.max(np.abs(x - XR[k + 1]) / scale[k + 1]))
    print(f"10-day RK4 n_sub 240 window {win}: max scaled err {worst:.3e}, refined sub-steps {nst}, env-steps with extra attempts {fl}", flush=True)
gt = np.load("tests/golden/step_tight.npz")
X, U, D, P, XT = gt["X"], gt["U"], gt["D"], gt["P"].astype(np.float64), gt["X_tight"]
sc = np.maximum(np.abs(XT), 1e-3 * np.abs(XT).max(axis=0, keepdims=True))
for win in (3, 4, 5, 6):
    Y = np.array([O.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, 240, 4, win)[0] for i in range(len(X))])
    e = np.abs(Y - XT) / sc
    i, j = np.unravel_index(np.argmax(e), e.shape)
    e2 = e.copy(); e2[i] = 0
    print(f"tight one-step tuples window {win}: max {e.max():.2e} (tuple {i} state {j}); all other tuples {e2.max():.2e}")
for nsub in (240, 480, 560, 640, 720):
    Y = np.array([O.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, nsub, 4, 4)[0] for i in range(len(X))])
    print(f"tight tuples, window 4, n_sub {nsub}: {np.max(np.abs(Y - XT) / sc):.2e}")
for name in ("step_tight_storm", "step_tight_jump"):
    g = np.load(f"tests/golden/{name}.npz")
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    sc = np.maximum(np.abs(XT), 1e-3 * np.abs(XT).max(axis=0, keepdims=True))
    for win in (3, 4):
        res = [O.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, 240, 4, win, verify=True) for i in range(len(X))]
        Y = np.array([r[0] for r in res]); failed = sum(bool(r[3]) for r in res)
        e = (np.abs(Y - XT) / sc).max(axis=1)
        print(f"{name} verified, window {win}: max {e.max():.2e}, above 1e-4: {(e > 1e-4).sum()}, failed {failed}", flush=True)
