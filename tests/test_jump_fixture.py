"""Raw-control-jump fixture on the CPU: tests/golden/step_tight_jump.npz (make_golden.py g_jump) holds 576
one-step maps "vents slammed open, screens pulled, cold, 8-40 m/s wind" / all-actuator corner flips with Radau-1e-11 truth
and the BDF-1e-6 solution (CVODES-tolerance proxy).  Checked here: the fixture itself, the oracle's restatement of the
step-doubling-verified guard, and the PRODUCT's gl_model.hpp arithmetic (host instantiation, tests/hostmath) on the hard
tuples.  The GPU kernels are checked through the C ABI in tests/test_gpu_jump.py."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np

COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5,
                   2.6e5, 6e4, 3.2e3, 60.])


def sce(a, b):
    """per-state scaled error of oracle/studies/stress_jump.py: |a - b| / max(|b|, 1e-3 x the state's typical magnitude)"""
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)


def judge(got, truth, abs_floor=1e-4):
    """-> (n above 1e-4 that are real, n at the metric's floor: a temperature within 1e4 x abs_floor of 0 C that is off by less
    than abs_floor kelvin -- 1e-4 K (|T| < 1 C) for fp64; the fp32 kernels: 2e-4 K (|T| < 2 C), which is what the 1.5e4 refined
    fp32 sub-steps of a pinned wet cover accumulate in rounding (tuple 8, Bogacki-Shampine: tAir -1.381 C off by 1.6e-4 K, the
    same kernel in fp64: 3e-6))"""
    e = sce(got, truth)
    bad = e > 1e-4
    floor = bad & (np.abs(got - truth) < abs_floor) & (np.arange(28)[None, :] < 22) & (np.abs(truth) < 1e4 * abs_floor)
    return int((bad & ~floor).any(axis=1).sum()), int(floor.any(axis=1).sum())


def test_fixture_is_what_it_says(golden):
    g = golden("step_tight_jump")
    assert len(g["X"]) >= 512 and g["seed"][0] == -1 and g["seed"][1] == -2          # the review's tuples A, B first
    assert np.all(g["truth_agreement"] < 2e-7)                                        # Radau 1e-11 ^ RK4-32 768
    assert abs(g["D"][0][4] - 12.860354734974642) < 1e-12 and abs(g["D"][1][4] - 25.542645227113603) < 1e-12
    assert np.all(np.isfinite(g["X_bdf"]))                                            # the CVODES proxy fails on none
    wrong, floor = judge(g["X_bdf"], g["X_tight"])
    assert wrong <= 3                                   # BDF at 1e-6 itself: a few 1e-4 ... 2e-3 (its tolerance), nothing gross
    assert sce(g["X_bdf"], g["X_tight"]).max() < 1e-2


import pytest


@pytest.mark.parametrize("n_sub,order,win", [(128, 5, 2), (240, 4, 4)])
def test_oracle_verified_guard_against_the_jump_truth(golden, oracle, n_sub, order, win):
    g = golden("step_tight_jump")
    p = golden("params_default")["p"].astype(np.float64)
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]

    def run(i):
        return oracle.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, n_sub, order, win, verify=True)
    with ThreadPoolExecutor(8) as ex:
        R = list(ex.map(run, range(len(X))))
    got = np.array([r[0] for r in R])
    assert not any(r[3] for r in R)                                                   # no failed integration
    wrong, floor = judge(got, XT)
    print(f"oracle order {order} n_sub {n_sub} verified on {len(X)} jump tuples: above 1e-4: {wrong} (+ {floor} at the metric floor), "
          f"max {sce(got, XT).max():.1e}, attempts beyond the first: {sum(r[1] for r in R)}")
    assert wrong == 0 and floor <= 3
    # the review's tuples A and B under the UNVERIFIED guard (the action path's integration): round 2 capped the refinement at
    # 16x, went unstable on the pinned cover and returned the wrong branch with failed = 0; now the sub-step follows the rate
    # bound down to 1/64 of the nominal one, the attempt is 'heavy' (>= 3x the nominal sub-steps) and gets verified by 2x
    for i in (0, 1):
        y, retries, refined, failed = oracle.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, n_sub, order, win)
        assert not failed and retries >= 1 and refined > 3 * n_sub and judge(y[None], XT[i][None])[0] == 0


def test_product_arithmetic_on_the_hard_jump_tuples(golden, oracle, hostmath):
    """gl_model.hpp's rk4_delta_guarded (host build) == the oracle's restatement on the tuples where round 2 was wrong; fp32
    within the bar of the truth."""
    g = golden("step_tight_jump")
    p = golden("params_default")["p"].astype(np.float64)
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    for i in list(range(0, 15)) + [40, 100, 300]:
        for (n, o, w) in ((128, 5, 2), (240, 4, 4), (270, 3, 3)):
            a = hostmath.step_guarded(X[i], U[i], D[i], p, False, 900.0, n, o, w, verify=True)
            b = oracle.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, n, o, w, verify=True)
            assert a[1] == b[1] and a[3] == b[3] and not a[3], (i, n, a[1:], b[1:])
            assert sce(a[0], b[0]).max() < 1e-7, (i, n, sce(a[0], b[0]).max())      # kinks amplify rounding: 2e-8 seen
        for (n, o, w) in ((128, 5, 2), (240, 4, 4)):
            y32 = hostmath.step_guarded(X[i], U[i], D[i], p, True, 900.0, n, o, w, verify=True)
            wrong, floor = judge(y32[0][None], XT[i][None], 2e-4)
            assert not y32[3] and wrong == 0, (i, n, sce(y32[0], XT[i]).max())


def test_c_bdf_stand_in_for_cvodes(golden, oracle):
    """oracle/gl_oracle.c gl_oracle_bdf (variable-order BDF, modified Newton, reused finite-difference Jacobian; rtol = atol =
    1e-6 -- the algorithm family and tolerances of the reference's CVODES call, greenlight_model.cpp:46-63): what bench.py times
    as `cpu_baseline`.  Against the tight truth it must sit where scipy's BDF at the same tolerances sits (the fixtures hold
    that solution), at a comparable number of right-hand sides."""
    p = golden("params_default")["p"].astype(np.float64)
    g = golden("step_tight_jump")
    n = 160
    got, nfev = oracle.bdf_batch(g["X"][:n], g["U"][:n], g["D"][:n], p)
    assert np.all(np.isfinite(got))
    e_c, e_scipy = sce(got, g["X_tight"][:n]).max(axis=1), sce(g["X_bdf"][:n], g["X_tight"][:n]).max(axis=1)
    print(f"C BDF on {n} jump tuples: max {e_c.max():.1e} median {np.median(e_c):.1e}, {nfev / n:.0f} RHS evaluations per env-step; "
          f"scipy BDF: max {e_scipy.max():.1e} median {np.median(e_scipy):.1e}")
    assert e_c.max() < 3 * max(e_scipy.max(), 2e-5) and np.median(e_c) < 3 * np.median(e_scipy) and nfev / n < 800
    t = golden("step_tight")
    for i in range(0, len(t["X"]), 5):
        y, nf, st = oracle.bdf(t["X"][i], t["U"][i], t["D"][i], t["P"][i])
        assert sce(y, t["X_tight"][i]).max() < 1e-4 and nf < 3 * t["nfev_bdf1e6"][i] + 100, (i, nf)
