"""CPU tests of the PRODUCT's arithmetic (greenlight-gym2_amd/csrc/gl_model.hpp) through a tests-only host build
(tests/hostmath/).  Same header the gfx950 kernels compile; here g++ instantiates it for float and double so the
tier-1/2/3 split, the fp32 reformulations and the delta-form RK4 are checked against the golden vectors without a GPU.
"""
import numpy as np

from conftest import scaled_err


def test_rhs_fp64_and_fp32_against_reference_text_vectors(hostmath, golden):
    g = golden("rhs_kat")
    X, U, D, P, DX = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["DX"]
    sc = np.maximum(np.abs(DX).max(axis=0), 1e-30)
    for per_env in (False, True):
        e64 = max(np.max(np.abs(hostmath.rhs(X[i], U[i], D[i], P[i], False, per_env) - DX[i]) / sc) for i in range(256))
        e32 = max(np.max(np.abs(hostmath.rhs(X[i], U[i], D[i], P[i], True, per_env) - DX[i]) / sc) for i in range(256))
        assert e64 < 1e-11, e64          # algebraically identical, different evaluation order
        assert e32 < 1e-4, e32


def test_step_map_against_oracle_and_tight(hostmath, oracle, golden):
    g = golden("step_tight")
    X, U, D, P, XT = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["X_tight"]
    ok = np.ones(len(X), dtype=bool)          # all tuples, incl. the harvest-switch zone (exact sub-flow)
    ref = np.array([oracle.rk4_split(X[i], U[i], D[i], P[i], 900.0, 256) for i in range(len(X))])
    g64 = np.array([hostmath.step(X[i], U[i], D[i], P[i], False) for i in range(len(X))])
    g32 = np.array([hostmath.step(X[i], U[i], D[i], P[i], True) for i in range(len(X))])
    assert scaled_err(g64[ok], ref[ok]) < 1e-9
    assert scaled_err(g32[ok], ref[ok]) < 2e-5
    assert scaled_err(g64[ok], XT[ok]) < 1.3e-5


def test_fp32_10day_rollout_meets_1e_4(hostmath, golden):
    """The north-star accuracy bar, evaluated on the product's own fp32 arithmetic (host instantiation)."""
    g = golden("rollout_10day")
    acts, w, XR = g["actions"], g["weather"], g["X"]
    p = golden("params_default")["p"].astype(np.float64)
    for f32, tol in ((False, 5e-6), (True, 1e-4)):
        x, u, X = XR[0].copy(), np.zeros(6), [XR[0]]
        for k in range(961):
            u = np.clip(u + acts[k] * np.float32(0.1), 0, 1)
            x = hostmath.step(np.float32(x) if f32 else x, np.float32(u) if f32 else u,
                              np.float32(w[k]) if f32 else w[k], p, f32)
            X.append(x)
        assert scaled_err(np.array(X), XR) < tol
