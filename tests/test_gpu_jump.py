"""Raw-control-jump fixture through the C ABI on the GPU: glgym_evalF and glgym_step(control = ...), the three
schemes, fp32 and fp64, against the TIGHT truth of tests/golden/step_tight_jump.npz (Radau 1e-11 ^ RK4-32 768; make_golden.py
g_jump) -- 576 one-step maps "vents slammed to 1, screens pulled to 0, cold, 8-40 m/s wind" and all-actuator corner flips,
incl. the review's tuples A and B, on which the round-2 kernels returned the wet cover on the wrong branch with failed = 0.

Bar: NO silent error above 1e-4 (per-state scaled error of oracle/studies/stress_jump.py; a temperature within 0.1 C of 0 C that
is off by < 1e-4 K counts as the metric's floor and is bounded separately), and a failed integration only where the BDF-1e-6
proxy of the reference's CVODES also fails (it fails on none of the fixture)."""
import numpy as np
import pytest

from test_jump_fixture import judge, sce

pytestmark = pytest.mark.gpu

SCHEMES = [("ls5", 128), ("rk4", 240), ("rk3", 270), ("rk2", 336)]


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("scheme,n_sub", SCHEMES)
def test_jump_step_maps_through_evalF(golden, scheme, n_sub, dtype):
    from gl_gym_amd import GreenLight
    g = golden("step_tight_jump")
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, n_sub=n_sub)
    got = m.evalF_batch(X, U, D)                     # raises GlgymOdeError on a failed row: none may fail (BDF-1e-6 does not)
    assert np.all(np.isfinite(got))
    wrong, floor = judge(got, XT, 1e-4 if dtype == "float64" else 2e-4)
    print(f"jump evalF {scheme} {dtype}: above 1e-4: {wrong} (+ {floor} at the metric floor), max {sce(got, XT).max():.1e}; "
          f"tuple A {sce(got[0], XT[0]).max():.1e}, tuple B {sce(got[1], XT[1]).max():.1e}")
    e = sce(got, XT)
    for i in np.nonzero((e > 1e-4).any(axis=1))[0]:
        j = int(e[i].argmax())
        print(f"   tuple {i} seed {g['seed'][i]} state {j} err {e[i, j]:.2e} abs {abs(got[i, j] - XT[i, j]):.2e} truth {XT[i, j]:.4f}")
    assert wrong == 0, (scheme, dtype)
    assert floor <= (12 if scheme == "rk2" else 6)
    m.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("scheme,n_sub", SCHEMES)
def test_jump_step_maps_through_step_kernel(golden, scheme, n_sub, dtype):
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("step_tight_jump")
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    B = len(X)
    w = np.repeat(D, 4, axis=0)                                  # env b integrates over row 4 b
    env = TomatoVecEnv(B, weather=w, dtype=dtype, scheme=scheme, n_sub=n_sub, season_length=0.02, pred_horizon=0,
                       auto_reset=False)
    env.reset()
    env.w_off_t.copy_(torch.arange(B, dtype=torch.int32, device=env.device) * 4)
    env.x.copy_(torch.as_tensor(X, dtype=env.tdtype, device=env.device))
    env.metrics_t.zero_()
    obs, r, done, infos = env.step_raw_control(U)                # step_raw_control: no delta-u clip (tomato_env.py:148-149)
    got = env.x.double().cpu().numpy()
    m = env.metrics()
    wrong, floor = judge(got, XT, 1e-4 if dtype == "float64" else 2e-4)
    print(f"jump step {scheme} {dtype}: above 1e-4: {wrong} (+ {floor} floor), max {sce(got, XT).max():.1e}; refined sub-steps "
          f"{m['n_refined_substeps']:.0f}, extra attempts {m['n_guard_retries']:.0f}, failed {m['n_ode_fail']:.0f}")
    assert m["n_ode_fail"] == 0 and not done.any()
    assert wrong == 0, (scheme, dtype)
    assert m["n_guard_retries"] >= B                             # verified mode: every env-step took at least two attempts
    env.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("scheme,n_sub", [("ls5", 128), ("rk4", 240)])
def test_two_rungs_at_a_time_equals_the_sequential_ladder(golden, scheme, n_sub, dtype):
    """Verified evalF calls on small batches integrate n_sub and 2 n_sub (then 4 n_sub and 8 n_sub) side by side on two lane groups per
    row (include/glgym.h glgym_set_ladder_parallel): the accepted attempt and the state must be the sequential ladder's BIT FOR BIT
    -- on the fixture whose tuples need the ladder most (every row takes two attempts, some three or four)."""
    from gl_gym_amd import GreenLight
    g = golden("step_tight_jump")
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, n_sub=n_sub)
    par = m.evalF_batch(X, U, D)
    m.set_ladder_parallel(False)
    seq = m.evalF_batch(X, U, D)
    assert np.array_equal(par, seq), (scheme, dtype, float(np.nanmax(np.abs(par - seq))))
    one_row = m.evalF_batch(X[5:6], U[5:6], D[5:6])           # ... and a row's result does not depend on the batch around it
    m.set_ladder_parallel(True)
    assert np.array_equal(m.evalF_batch(X[5:6], U[5:6], D[5:6]), one_row) and np.array_equal(one_row[0], seq[5])
    # row by row (one wavefront, 56 of its lanes idle), other kernels in between: the first build of this path left the row-0 lane of
    # the accepted lane group unwritten depending on what earlier kernels had left in a register (hipcc 7.2; gl_model_quad.hpp)
    other = GreenLight(28, 6, 10, 208, 900.0, dtype="float64" if dtype == "float32" else "float32", scheme=scheme, n_sub=n_sub)
    for i in range(0, 96, 2):
        other.evalF_batch(X[i:i + 3], U[i:i + 3], D[i:i + 3])
        got = m.evalF_batch(X[i:i + 1], U[i:i + 1], D[i:i + 1])[0]
        assert np.array_equal(got, seq[i]), (scheme, dtype, i, np.nonzero(got != seq[i])[0])
    other.close()
    if dtype == "float32":                                     # the one-lane fp32 evalF kernel (what batches beyond 16 384 rows run)
        m.set_layout("one")
        one = m.evalF_batch(X, U, D)
        wrong, floor = judge(one, XT, 2e-4)
        assert wrong == 0 and floor <= 6, (scheme, wrong, floor)
    m.close()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("scheme,n_sub", SCHEMES)
def test_step_kernel_two_rungs_at_a_time_equals_the_sequential_ladder(golden, scheme, n_sub, dtype):
    """Round 6: verified glgym_step launches (raw controls) on batches of up to 8 192 environments run the ladder two rungs at a time on
    two lane groups per environment (`step_kernel_quad<..., PAIR>`, include/glgym.h glgym_set_ladder_parallel).  On the raw-jump fixture
    -- every environment takes two attempts, some three or four -- state, reward, done, info, the per-env step_flags word (first-attempt
    flags, extra attempts, sub-steps beyond nominal, how the result was accepted) and the metric accumulators must be the sequential
    ladder's BIT FOR BIT; and a ragged batch (B = 5: one wavefront with 24 idle lanes) must not depend on the lanes around it."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("step_tight_jump")
    X, U, D = g["X"], g["U"], g["D"]
    out = {}
    for par in (True, False):
        for B in (len(X), 5):
            w = np.repeat(D[:B], 4, axis=0)
            env = TomatoVecEnv(B, weather=w, dtype=dtype, scheme=scheme, n_sub=n_sub, season_length=0.02, pred_horizon=0, auto_reset=False)
            env.set_ladder_parallel(par)
            env.reset()
            env.w_off_t.copy_(torch.arange(B, dtype=torch.int32, device=env.device) * 4)
            env.x.copy_(torch.as_tensor(X[:B], dtype=env.tdtype, device=env.device))
            env.metrics_t.zero_()
            obs, r, done, info = env.step_raw_control(U[:B])
            m = env.metrics()
            out[(par, B)] = (env.x.cpu().numpy().copy(), r.copy(), done.copy(), info.copy(), env.step_flags_t.cpu().numpy().copy(), env.u.cpu().numpy().copy(),
                             env.timestep_t.cpu().numpy().copy(), {k: m[k] for k in ("n_ode_fail", "n_guard_retries", "n_refined_substeps", "n_flag_err", "n_flag_branch",
                                                                                    "n_flag_cap", "n_flag_heavy", "n_env_steps")})
            env.close()
    for B in (len(X), 5):
        a, b = out[(True, B)], out[(False, B)]
        for k in range(7):
            assert np.array_equal(a[k], b[k], equal_nan=True), (scheme, dtype, B, k, np.nonzero(a[k] != b[k]))
        assert a[7] == b[7], (a[7], b[7])
    assert np.array_equal(out[(True, 5)][0], out[(True, len(X))][0][:5])            # rows do not depend on the batch around them
    flags = out[(True, len(X))][4]
    assert np.all(((flags >> 8) & 0xff) >= 1) and out[(True, len(X))][7]["n_guard_retries"] >= len(X)       # verified: >= 1 extra attempt each
    print(f"step kernel, two rungs at a time == sequential ladder ({scheme} {dtype}): {len(X)} + 5 envs bit-identical; envs with 3+ attempts "
          f"{int((((flags >> 8) & 0xff) >= 2).sum())}, accepted by agreement although flagged {int(((flags & 32) != 0).sum())}, finest alone {int(((flags & 64) != 0).sum())}")


def test_unverified_mode_flags_what_round_2_missed(golden):
    """GLGYM_VERIFY_NEVER = the action path's integration (guard only).  On the review's tuples A and B the branch invariant now
    sends the env-step up the ladder: right, or failed -- not silently wrong."""
    from gl_gym_amd import GreenLight
    from gl_gym_amd._lib import GlgymOdeError
    g = golden("step_tight_jump")
    for dtype, scheme, n_sub in (("float64", "ls5", 128), ("float32", "ls5", 128), ("float64", "rk4", 240), ("float32", "rk4", 240)):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme=scheme, n_sub=n_sub)
        m.set_verify("never")
        for i in (0, 1):
            try:
                got = m.evalF_batch(g["X"][i:i + 1], g["U"][i:i + 1], g["D"][i:i + 1])
            except GlgymOdeError:
                continue
            wrong, floor = judge(got, g["X_tight"][i:i + 1], 2e-4)
            assert wrong == 0, (dtype, scheme, i, sce(got[0], g["X_tight"][i]).max())
        m.close()
