import sys, numpy as np
sys.path.insert(0, "greenlight-gym2_amd"); sys.path.insert(0, "tests")
from gl_gym_amd import GreenLight
from test_gpu_fuzz import _tuples
golden = lambda name: np.load(f"tests/golden/{name}.npz")
N = 60
X, U, D, P = _tuples(N, golden)
scale = np.maximum(np.abs(X).max(axis=0), 1e-3)
order = sys.argv[1].split(",")
NS = {"rk4": 256, "ls5": 128, "rk3": 282, "rk2": 360}
for scheme in order:
    n_sub = NS[scheme]
    res = {}
    for label, par, lay, withp in (("pair+p", True, "auto", True), ("pair", True, "auto", False), ("seq+p", False, "auto", True), ("seq", False, "auto", False)):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype="float32", scheme=scheme, n_sub=n_sub)
        m.set_ladder_parallel(par); m.set_layout(lay)
        if withp:
            got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(N)])
        else:
            got = np.array([m.evalF_batch(X[i:i+1], U[i:i+1], D[i:i+1])[0] for i in range(N)])
        res[label] = got
        m.close()
    e = lambda a, b: float(np.nanmax(np.abs(res[a] - res[b]) / np.maximum(np.abs(res[b]), scale)))
    nodef = [i for i in range(N) if i % 3 != 0]
    print(scheme, "pair+p vs seq+p", e("pair+p", "seq+p"), "| pair vs seq", e("pair", "seq"), "| default-p rows, pair+p vs pair:",
          float(np.nanmax(np.abs(res["pair+p"][nodef] - res["pair"][nodef]) / np.maximum(np.abs(res["pair"][nodef]), scale))),
          "| first bad row", next((i for i in range(N) if np.nanmax(np.abs(res["pair+p"][i] - res["seq+p"][i]) / np.maximum(np.abs(res["seq+p"][i]), scale)) > 1e-3), None))
    print("    NaN rows: pair+p", int(np.isnan(res["pair+p"]).any(axis=1).sum()), "pair", int(np.isnan(res["pair"]).any(axis=1).sum()), "seq", int(np.isnan(res["seq"]).any(axis=1).sum()))
