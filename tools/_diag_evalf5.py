import os, sys, numpy as np
sys.path.insert(0, "greenlight-gym2_amd"); sys.path.insert(0, "tests")
from gl_gym_amd import GreenLight
from test_gpu_fuzz import _tuples
golden = lambda name: np.load(f"tests/golden/{name}.npz")
N = 30
X, U, D, P = _tuples(N, golden)
for scheme, n_sub in (("rk4", 256), ("rk3", 282), ("ls5", 128), ("rk2", 360)):
    for dtype in ("float32", "float64"):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, n_sub=n_sub)
        pat = {}
        for i in range(N):
            got = m.evalF_batch(X[i:i+1], U[i:i+1], D[i:i+1])[0]
            key = tuple(np.nonzero(np.isnan(got))[0].tolist())
            pat[key] = pat.get(key, 0) + 1
        print(scheme, dtype, "NaN column patterns:", pat)
        m.close()
