import sys, numpy as np
sys.path.insert(0, "greenlight-gym2_amd"); sys.path.insert(0, "tests")
from gl_gym_amd import GreenLight
from test_gpu_fuzz import _tuples
golden = lambda name: np.load(f"tests/golden/{name}.npz")
N = 400
X, U, D, P = _tuples(N, golden)
scale = np.maximum(np.abs(X).max(axis=0), 1e-3)
for scheme, n_sub in (("rk4", 256), ("ls5", 128), ("rk3", 282)):
    ref = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, n_sub=n_sub)
    r64 = np.array([ref.evalF(X[i], U[i], D[i], P[i]) for i in range(N)]); ref.close()
    for label, par, lay in (("pair", True, "auto"), ("seqquad", False, "auto"), ("one", False, "one")):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype="float32", scheme=scheme, n_sub=n_sub)
        m.set_ladder_parallel(par); m.set_layout(lay)
        got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(N)])
        e = np.abs(got - r64) / np.maximum(np.abs(r64), scale)
        bad = np.nonzero(e.max(axis=1) > 1e-4)[0]
        print(scheme, label, "max", e.max(), "bad rows", bad[:10], [int(e[b].argmax()) for b in bad[:10]])
        m.close()
