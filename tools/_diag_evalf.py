import sys, numpy as np
sys.path.insert(0, "greenlight-gym2_amd")
from gl_gym_amd import GreenLight
g = np.load("tests/golden/step_tight_jump.npz")
X, U, D = g["X"][:64], g["U"][:64], g["D"][:64]
for scheme, n_sub in (("ls5", 128), ("rk4", 256), ("rk3", 282), ("rk2", 360)):
    ref = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, n_sub=n_sub)
    r64 = ref.evalF_batch(X, U, D); ref.set_ladder_parallel(False); r64s = ref.evalF_batch(X, U, D); ref.close()
    sc = np.maximum(np.abs(r64).max(axis=0), 1e-3)
    print(scheme, "fp64 pair vs seq", np.nanmax(np.abs(r64 - r64s) / sc))
    for verify in ("auto", "never"):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype="float32", scheme=scheme, n_sub=n_sub)
        m.set_verify(verify)
        out = {}
        for label, par, lay in (("pair", True, "auto"), ("seqquad", False, "auto"), ("one", False, "one")):
            m.set_ladder_parallel(par); m.set_layout(lay)
            try:
                out[label] = m.evalF_batch(X, U, D)
            except Exception as e:
                print("   ", label, "raised", str(e)[:80]); out[label] = np.full_like(r64, np.nan)
        print(scheme, verify, {k: float(np.nanmax(np.abs(v - r64) / sc)) for k, v in out.items()}, "rows bad (pair):", np.nonzero((np.abs(out["pair"] - r64) / sc).max(axis=1) > 1e-3)[0][:10])
        m.close()
