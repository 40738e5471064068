"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed golden fixtures.

Tolerances (north_star: <= 1e-4 relative state error over a 10-day rollout, fp64 -> fp32):
  fp64 kernels vs oracle RHS / RK4      : 1e-9 scaled   (different but algebraically identical expression order)
  fp32 kernels, one env-step            : 2e-5 scaled vs fp64 oracle RK4
  fp32 kernels, 10-day / 961-step rollout: 1e-4 scaled vs the tight (Radau 1e-11) fixture  <- the headline bar
"""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _scheme_arithmetic_not_the_verified_ladder():
    """This module checks the kernels' SCHEME at a given n_sub against the oracle's restatement of that scheme and against the
    fixtures, also through glgym_evalF / step_raw_control -- entry points that by default integrate step-doubling VERIFIED
    (they would return the 2 n_sub attempt; include/glgym.h glgym_verify).  GLGYM_VERIFY=never makes every handle created
    here start unverified; the verified ladder has its own tests (test_gpu_jump.py, test_gpu_storm.py, test_gpu_fuzz.py)."""
    import os
    old = os.environ.get("GLGYM_VERIFY")
    os.environ["GLGYM_VERIFY"] = "never"
    yield
    if old is None:
        os.environ.pop("GLGYM_VERIFY", None)
    else:
        os.environ["GLGYM_VERIFY"] = old


@pytest.fixture(scope="module")
def models():
    from gl_gym_amd import GreenLight
    m64 = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme="rk4", n_sub=256)      # the oracle references below use RK4 at 256
    m32 = GreenLight(28, 6, 10, 208, 900.0, dtype="float32", scheme="rk4", n_sub=256)
    yield m64, m32
    m64.close(); m32.close()


def test_rhs_matches_reference_text_vectors(models, golden):
    g = golden("rhs_kat")
    X, U, D, P, DX = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["DX"]
    sc = np.maximum(np.abs(DX).max(axis=0), 1e-30)
    m64, m32 = models
    default = np.all(P == P[-1], axis=1)          # rows with the default parameter block
    m64.set_params(P[-1]); m32.set_params(P[-1])
    e64 = np.abs(m64.rhs(X[default], U[default], D[default]) - DX[default]) / sc
    e32 = np.abs(m32.rhs(X[default], U[default], D[default]) - DX[default]) / sc
    assert e64.max() < 1e-11, e64.max()
    assert e32.max() < 2e-4, e32.max()


def test_evalF_signature_and_value(models, golden, oracle):
    g = golden("step_tight")
    X, U, D, P, XT = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["X_tight"]
    m64, m32 = models
    out = m64.evalF(X[0], U[0], D[0], P[0])
    assert isinstance(out, list) and len(out) == 28 and all(isinstance(v, float) for v in out)
    ok = np.ones(len(X), dtype=bool)          # every tuple, incl. the harvest-switch zone (exact sub-flow)
    ref = np.array([oracle.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, 256, 4, 4)[0] for i in range(len(X))])
    got64 = np.array([m64.evalF(X[i], U[i], D[i], P[i]) for i in range(len(X))])
    got32 = np.array([m32.evalF(X[i], U[i], D[i], P[i]) for i in range(len(X))])
    assert scaled_err(got64[ok], ref[ok]) < 1e-9
    assert scaled_err(got32[ok], ref[ok]) < 2e-5
    # vs the tight stiff solve on perturbed (off-equilibrium) tuples; the CVODES-tolerance proxy (BDF rtol=atol=1e-6) sits
    # at 1.3e-5 on the same tuples.  Round 4 (cover conduction exact, nominal sub-step 3.5 - 3.75 s, four-sub-step tier-2b
    # window of 14 - 15 s): 5.4e-5 / 6.2e-5 -- one artificial tuple that starts with an empty carbohydrate buffer carries it
    # (cBuf 0 -> 274 mg in the step, 0.02 mg off; every other state of every tuple < 2.4e-5, asserted below) -- inside the 1e-4 bar
    assert scaled_err(got64[ok], XT[ok]) < 5.6e-5
    rest = np.array([i for i in range(len(X)) if i != 61])
    assert scaled_err(got64[rest], XT[rest]) < 2.5e-5
    from gl_gym_amd import GreenLight
    e_bdf = scaled_err(g["X_bdf1e6"], XT)          # a BDF solve at the reference's tolerances (rtol = atol = 1e-6) on the same tuples: 1.3e-5
    # --- the reference-compatible class as a maintainer gets it (no keyword): the PARITY preset -- the five-stage 2N scheme at n_sub 192
    # with one sub-step per tier-2b window -- must sit INSIDE the band the reference solver's tolerances keep from the tight solution
    m_def = GreenLight(28, 6, 10, 208, 900.0)
    assert (m_def.scheme, m_def.n_sub, m_def.window, m_def.preset) == ("ls5", 192, 1, "parity")
    e_def = scaled_err(m_def.evalF_batch(X, U, D, P), XT)
    m_def.close()
    print(f"fp64 vs tight one-step solutions: RK4 n_sub 256 {scaled_err(got64[ok], XT[ok]):.2e}; GreenLight() default = ls5 parity preset {e_def:.2e} "
          f"(BDF rtol = atol = 1e-6: {e_bdf:.2e})")
    assert e_def < 1.3e-5 and e_def <= e_bdf
    # --- the THROUGHPUT preset of both fourth-order schemes (what the batched envs and bench.py's `value` run): inside the 1e-4 bar with
    # the one artificial tuple at 6.1e-5, every other tuple well below
    for scheme, n_expect, rest_tol in (("ls5", 128, 4.0e-5), ("rk4", 240, 2.5e-5)):
        m_thr = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, preset="throughput")
        assert m_thr.n_sub == n_expect and m_thr.window == 0
        got = m_thr.evalF_batch(X, U, D, P)
        m_thr.close()
        print(f"fp64 {scheme} throughput preset n_sub {n_expect}: {scaled_err(got, XT):.2e}, without tuple 61 {scaled_err(got[rest], XT[rest]):.2e}")
        assert scaled_err(got, XT) < 6.3e-5 and scaled_err(got[rest], XT[rest]) < rest_tol
    # --- the parity configuration of classical RK4 (include/glgym.h): n_sub 640
    m_par = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme="rk4")
    assert m_par.n_sub == 640
    e_par = scaled_err(m_par.evalF_batch(X, U, D, P), XT)
    m_par.close()
    print(f"fp64 RK4 parity configuration n_sub 640: {e_par:.2e}")
    assert e_par < 1.3e-5
    # batched call with per-row crop parameters == row-by-row calls
    got_b = m64.evalF_batch(X[:16], U[:16], D[:16], P[:16])
    assert scaled_err(got_b, got64[:16]) < 1e-12


def test_n_sub_4_is_refined_to_what_the_ode_needs_or_flagged(golden, oracle):
    """BASELINE config 3 asks for 'RK4 with 4 sub-steps': the ODE is stiff (lambda_max ~ 0.67 1/s), a fixed step of 225 s
    overflows at once.  n_sub is the NOMINAL count: the stability control inserts the >= 224 sub-steps the fast block needs (since
    round 3 down to 1/64 of the nominal sub-step), such an attempt is 'heavy' and gets verified by step doubling.  What n_sub = 4
    then still sets is the length of the slow tier's windows (450 s), which costs accuracy: 3e-4 against a fine solve, outside
    the 1e-4 bar that n_sub = 320 meets -- or the env-step is flagged like a failed CVODES call (done = 1, state unchanged).
    Never a non-finite or silently wild state."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    env = TomatoVecEnv(64, weather=w, dtype="float32", scheme="rk4", n_sub=4, season_length=1, auto_reset=False)
    env.reset()
    x_before = env.x.double().cpu().numpy().copy()
    obs, r, done, info = env.step(np.zeros((64, 6), np.float32))
    x_after = env.x.double().cpu().numpy()
    m = env.metrics()
    assert np.all(np.isfinite(x_after))
    truth = oracle.rk4(x_before[0], np.zeros(6), w[0], env.p.astype(np.float64), 900.0, 16384)
    for b in range(64):
        if done[b]:
            assert np.array_equal(x_after[b], x_before[b])
        else:
            assert scaled_err(x_after[b][None], truth[None]) < 1e-3, b
    assert m["n_ode_fail"] == done.sum()
    assert m["n_refined_substeps"] >= 64 * 220 or done.all()        # the stability floor was inserted, per lane
    print(f"n_sub 4: {int(done.sum())} of 64 flagged; others {scaled_err(x_after[~done], np.repeat(truth[None], (~done).sum(), 0)) if (~done).any() else 0:.1e} "
          f"from the fine solve with {m['n_refined_substeps'] / 64:.0f} inserted sub-steps per env")
    env.close()


@pytest.mark.parametrize("scheme,preset", [("ls5", "throughput"), ("ls5", "parity"), ("rk4", "throughput"), ("rk4", "parity"), ("rk2", "throughput"),
                                           ("rk3", "throughput")])
@pytest.mark.parametrize("fixture,dtype,tol", [("rollout_10day", "float64", 5e-6), ("rollout_10day", "float32", 1e-4),
                                               ("rollout_3day_synth", "float64", 5e-6),
                                               ("rollout_3day_synth", "float32", 1e-4)])
def test_10day_rollout_vs_tight_fixture(golden, fixture, dtype, tol, scheme, preset):
    """The headline accuracy bar: step() with the fixture's action sequence vs the tight (Radau 1e-11) states.
    rollout_10day: 961 steps on Bleiswijk autumn weather; rollout_3day_synth: 289 steps on midsummer-like synthetic
    weather (670 W/m2 peaks, strong photosynthesis and ventilation, air temperature 6..28 C)."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden(fixture)
    acts, w, XR = g["actions"], g["weather"], g["X"]
    B = 64                                    # 64 identical envs: also checks lane-independence
    n_steps = len(acts)
    if dtype == "float64":
        # THROUGHPUT presets: the midpoint rule at n_sub = 336: second order, 7.6e-6 / 9.6e-6 in fp64; RK4 at 240 and the five-stage 2N
        # scheme at 128 (15 s / 14 s tier-2b windows): 1.5e-5 / 1.4e-5; the three-stage scheme at 270 (10 s windows): 6.8e-6 / 6.6e-6.
        # PARITY presets (ls5 192 / one-sub-step windows: 2.9e-6 / 2.7e-6; rk4 640): the round-3 bound of 5e-6 stays asserted
        tol = 5e-6 if preset == "parity" else (2e-5 if scheme in ("rk4", "ls5") else 1e-5)
    elif preset == "parity":
        tol = 4e-5                            # fp32 at the parity presets: rounding, not the scheme, sets it (2.5e-5 measured)
    env = TomatoVecEnv(B, weather=w, dtype=dtype, scheme=scheme, preset=preset, season_length=(n_steps - 1) // 96, pred_horizon=0.5,
                       auto_reset=False)
    env.reset()
    import torch
    X = [env.x[0].double().cpu().numpy()]
    for k in range(n_steps):
        a = torch.as_tensor(np.repeat(acts[k][None], B, 0), device=env.device)
        _, _, done, _ = env.step_tensor(a, want_obs=False)
        X.append(env.x[0].double().cpu().numpy())
    X = np.array(X)
    assert np.array_equal(env.x[0].cpu().numpy(), env.x[B - 1].cpu().numpy())
    err = scaled_err(X, XR)
    print(f"{fixture} {dtype} {scheme} {preset} (n_sub {env.n_sub}, window {env.window or 'scheme'}): max scaled rel err vs tight oracle = {err:.3e}")
    assert err < tol
    assert not bool(done[0]) or k == n_steps - 1        # no failed integration on the way (done only at the season's end)
    env.close()


def test_config3_shard_10day_horizon_at_full_batch(golden):
    """BASELINE configs[3] (batch 524 288 over 8 GPUs, fp32, 10-day horizon) is 65 536 environments per GPU with no data-path
    collective, so one rank's shard IS the configuration on the device side: 65 536 environments through the
    961 steps of the 10-day fixture (one-lane-per-environment kernel, the headline's), every environment within the 1e-4 bar of
    the tight (Radau 1e-11) states all the way, no failed integration, and -- same inputs in every lane -- every row bit-identical
    to row 0 (lane-, wave- and workgroup-independence at size)."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("rollout_10day")
    acts, w, XR = g["actions"], g["weather"], g["X"]
    B, n_steps = 65536, len(acts)
    env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=(n_steps - 1) // 96, pred_horizon=0.5, auto_reset=False)
    env.reset()
    a_all = torch.as_tensor(acts, device=env.device)
    worst = 0.0
    for k in range(n_steps):
        env.step_tensor(a_all[k][None].expand(B, 6).contiguous(), want_obs=False)
        if k % 48 == 47 or k == n_steps - 1:
            rows = env.x[[0, 777, 40000, B - 1]].double().cpu().numpy()
            worst = max(worst, max(scaled_err(r[None], XR[k + 1][None]) for r in rows))
    x = env.x
    assert bool((x == x[0:1]).all())
    m = env.metrics()
    print(f"config 3 shard: 65 536 envs x {n_steps} steps (10 days), fp32 {env.scheme} n_sub {env.n_sub}: max scaled err vs tight fixture {worst:.2e}; "
          f"failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}")
    assert worst < 1e-4 and m["n_ode_fail"] == 0
    env.close()


def test_config3_shard_distinct_environments_10day_against_fine_truth(oracle):
    """The same shard with 65 536 DISTINCT environments (the fixture test above runs identical rows): the bench
    workload's synthetic year, per-environment episode starts, jittered states, fresh U(-1, 1) actions every step, over the 10-day
    horizon (961 steps); eight environments spread over the wavefronts are integrated alongside on the CPU with plain classical RK4
    at 2 048 sub-steps from the same controls -- free-running, never re-synchronised -- and must stay inside the 1e-4 bar to the end."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    B, n_steps = 65536, 961
    w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
    starts = np.arange(0, 35040 - 5760 - 60, 96)
    env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=60, pred_horizon=0.5, seed=666, start_rows=starts, auto_reset=True)
    env.reset_tensor()
    dev = env.device
    env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)).to(env.tdtype))
    gen = torch.Generator(device=dev).manual_seed(667)
    pick = np.array([8192 * i + (11 * i) % 64 for i in range(8)])
    p = env.p.astype(np.float64)
    w_off = env.w_off_t.cpu().numpy()[pick]
    x_true = env.x[pick].double().cpu().numpy().copy()
    pool = ThreadPoolExecutor(8)
    worst = 0.0
    for k in range(n_steps):
        env.action_t.uniform_(-1.0, 1.0, generator=gen)
        env._launch_step(raw_control=False)
        u = env.u[pick].double().cpu().numpy()
        x_true = np.array(list(pool.map(lambda j: oracle.rk4_split(x_true[j], u[j], w[w_off[j] + k], p, 900.0, 2048), range(8))))
        if k % 96 == 95 or k == n_steps - 1:
            worst = max(worst, scaled_err(env.x[pick].double().cpu().numpy(), x_true))
        env._launch_reset(env.done_t)
    m = env.metrics()
    print(f"config 3 shard, 65 536 distinct envs x {n_steps} steps: 8 envs free-running vs RK4-2048 {worst:.1e}; failed {m['n_ode_fail']:.0f}, "
          f"extra attempts {m['n_guard_retries']:.0f}, refined sub-steps per env-step {m['n_refined_substeps'] / m['n_env_steps']:.3f}")
    assert worst < 1e-4 and m["n_ode_fail"] == 0 and m["n_env_steps"] == B * n_steps
    env.close()


@pytest.mark.parametrize("scheme,n_sub", [("ls5", 128), ("rk4", 256)])
def test_step_kernel_matches_env_oracle(golden, oracle, scheme, n_sub):
    """Fused step (control clip, weather row, sub-stepper, reward, info, terminal test) vs the numpy env oracle."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from oracle.gl_env_oracle import OracleTomatoEnv, INFO_KEYS
    w = golden("rollout_10day")["weather"]
    B = 96
    env = TomatoVecEnv(B, weather=w, dtype="float64", scheme=scheme, n_sub=n_sub, season_length=0.05, start_rows=[0, 40, 300],
                       start_days=[0.0, 0.4167, 3.125], seed=5, auto_reset=False)
    obs0 = env.reset()
    w_off = env.w_off_t.cpu().numpy(); sd = env.start_day_t.cpu().numpy()
    rng = np.random.default_rng(1)
    orcs = []
    for b in range(0, B, 6):
        o = OracleTomatoEnv(weather=w[w_off[b]:], p=env.p, season_length=0.05, integrator=scheme, n_sub=n_sub,
                            train_years=[0], train_days=[float(sd[b])], seed=0)
        ob = o.reset()
        assert np.allclose(ob, obs0[b], rtol=2e-6, atol=1e-6)
        orcs.append((b, o))
    for k in range(6):
        acts = rng.uniform(-1, 1, (B, 6)).astype(np.float32)
        obs, rew, dones, infos = env.step(acts)
        x = env.x.cpu().numpy()
        for b, o in orcs:
            ob, r, term, info = o.step(acts[b])
            assert scaled_err(x[b], o.x) < 1e-8
            assert abs(r - rew[b]) < 1e-6
            assert term == bool(dones[b])
            for key in INFO_KEYS:
                assert abs(info[key] - infos[b][key]) < 1e-6 * max(1.0, abs(info[key])), key
            assert np.allclose(ob, obs[b], rtol=2e-6, atol=2e-5), np.abs(ob - obs[b]).argmax()
    # season_length 0.05 d -> N = 4: terminal at the 5th step (N + 1 steps per episode)
    assert dones.all()
    env.close()


def test_batch_properties_at_full_size():
    """Size-independent properties at BASELINE's batch (65 536): permutation equivariance, determinism,
    metric accumulators == per-env sums."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    B = 65536
    w = synthetic_weather(n_rows=4000)
    starts = list(range(0, 2000, 37))
    env = TomatoVecEnv(B, weather=w, dtype="float32", n_sub=32, season_length=1, start_rows=starts, seed=11,
                       auto_reset=False)
    env.set_n_sub(224)
    env.reset()
    g = torch.Generator(device=env.device); g.manual_seed(0)
    a = torch.rand(B, 6, generator=g, device=env.device) * 2 - 1
    x0 = env.x_T.clone(); u0 = env.u_T.clone(); w0 = env.w_off_t.clone()
    env.step_tensor(a, want_obs=False)
    x1 = env.x_T.clone(); r1 = env.reward_t.clone()
    m = env.metrics()
    assert abs(m["sum_reward"] - float(r1[:B].double().sum())) < 1e-3 * B * 1e-2 + 1.0
    assert m["n_env_steps"] == B and m["n_ode_fail"] == 0
    # permute the envs, step again from the same state: outputs permute identically (bitwise)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(env.device)
    env.x_T[:, :B] = x0[:, :B][:, perm]; env.u_T[:, :B] = u0[:, :B][:, perm]
    env.w_off_t.copy_(w0[perm]); env.timestep_t.zero_()
    env.step_tensor(a[perm], want_obs=False)
    assert torch.equal(env.x_T[:, :B], x1[:, :B][:, perm])
    assert torch.equal(env.reward_t[:B], r1[:B][perm])
    assert torch.isfinite(env.x_T).all()
    env.close()


def test_generic_kernel_with_non_default_parameters(golden, oracle):
    """A parameter block that differs from the default one takes the generic step kernel (constants in SGPRs
    instead of compile-time literals) and exercises the branches the default block folds away: interlights on,
    FIR-transparent cover, grow-pipe emissivity, roof-only ventilation switched off."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    p = golden("params_default")["p"].astype(np.float64).copy()
    rng = np.random.default_rng(42)
    p[:127] *= 1 + 0.05 * rng.uniform(-1, 1, 127)
    p[70], p[67] = 0.05, 0.12                 # cover transmits some FIR -> sky terms live
    p[165] = 0.5                              # grow pipes radiate
    p[194], p[195], p[198] = 0.02, 0.8, 1.5   # interlight geometry present (their power stays 0)
    p[8] = 1.2                                # etaRoofThr > 1 -> the "else" ventilation branch
    for dtype, tol, scheme, n_sub, order, win in (("float64", 1e-8, "rk4", 256, 4, 4), ("float32", 5e-5, "rk4", 256, 4, 4),
                                                  ("float64", 1e-8, "ls5", 128, 5, 2), ("float32", 5e-5, "ls5", 128, 5, 2)):
        env = TomatoVecEnv(64, weather=w, params=p.astype(np.float32), dtype=dtype, scheme=scheme, n_sub=n_sub, season_length=1,
                           start_rows=[0, 50], seed=2, auto_reset=False)
        p32 = env.p.astype(np.float64)
        env.reset()
        w_off = env.w_off_t.cpu().numpy()
        acts = rng.uniform(-1, 1, (64, 6)).astype(np.float32)
        x = env.x.double().cpu().numpy().copy()
        for k in range(3):
            u_prev = env.u.double().cpu().numpy().copy()
            x_prev = env.x.double().cpu().numpy().copy()
            env.step(acts)
            xg = env.x.double().cpu().numpy()
            for b in range(0, 64, 9):
                u = np.clip(u_prev[b] + acts[b] * np.float32(0.1), 0, 1)
                ref = oracle.rk_sc_guarded(x_prev[b], u, w[w_off[b] + k], p32, 900.0, n_sub, order, win)[0]
                assert scaled_err(xg[b], ref) < tol, (dtype, scheme, k, b)
        env.close()


def test_config1_rule_based_day_against_fixture(golden):
    """BASELINE config 1 (rule-based controls, 1 day, Bleiswijk), batched on the GPU, against the fixture built from
    the reference's own controller and reward classes.

    The closed loop (rule-based controller at dt = 900 s) is chaotic: the controller bang-bangs and a 1e-7 state
    perturbation grows to O(1) within ~50 steps for ANY integrator (checked on the CPU oracle with n_sub = 1024), so
    free-running trajectories are not comparable.  Parity is therefore teacher-forced: at every step the env state
    is set to the fixture state, then (controls, next state, reward, info, obs) of that one step are compared."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.baseline import RuleBasedController
    from gl_gym_amd import INFO_KEYS
    g = golden("env_rulebased_1day")
    X, R, U, INFO, OBS = g["x"], g["reward"], g["u"], g["info"], g["obs"]
    keys = [str(k) for k in g["info_keys"]]
    ctrl = RuleBasedController()
    # (this module integrates UNVERIFIED, see the fixture at its top: the error is that of ONE attempt and is set by the tier-2b
    # window's length in seconds whatever the scheme -- CPU checker, same tuples: 3.5 s 3.2e-6, 1.76 s 7.9e-7, 4.7 s 5.6e-6, 15 s 5.8e-5)
    #   ls5 512 / window 1 (1.76 s)  the round-3 bound of 1e-6 for this day stays asserted
    #   rk4 1024 (window 4 = 3.5 s)  also checks info and observations of every step against the fixture
    #   ls5 parity preset (192 / 1)  and throughput preset (128 / 2), rk4 256
    for scheme, n_sub, window, tol in (("ls5", 512, 1, 1e-6), ("rk4", 1024, 0, 4e-6), ("ls5", 192, 1, 7e-6), ("ls5", 128, 0, 8e-5), ("rk4", 256, 0, 8e-5)):
        env = TomatoVecEnv(8, weather=g["weather"], params=g["p"], dtype="float64", scheme=scheme, n_sub=n_sub, window=window, season_length=1,
                           start_rows=[0], start_days=[0.0], auto_reset=False)
        obs = env.reset()
        np.testing.assert_allclose(obs[0], OBS[0], rtol=1e-6, atol=1e-5)
        worst_x = worst_r = worst_u = 0.0
        for k in range(97):
            env.x_T[:, :8] = torch.as_tensor(X[k], device=env.device)[:, None]
            u = env.rule_based_controls(ctrl)
            worst_u = max(worst_u, float(np.abs(u[0].cpu().numpy() - U[k]).max()))
            obs, rew, done, info = env.step_raw_control(U[k][None].repeat(8, 0))
            worst_x = max(worst_x, scaled_err(env.x[0].cpu().numpy(), X[k + 1]) if k else 0.0)
            sc = np.maximum(np.abs(X[k + 1]), 1e-3 * np.abs(X).max(axis=0))
            worst_x = max(worst_x, float(np.max(np.abs(env.x[0].cpu().numpy() - X[k + 1]) / sc)))
            worst_r = max(worst_r, abs(float(rew[0]) - R[k]))
            assert bool(done[0]) == (k == 96)                      # episode = N + 1 = 97 steps
            if tol <= 4e-6:                                        # the tight configurations: info and observations of every step too
                for j, key in enumerate(keys):
                    assert abs(info[INFO_KEYS.index(key), 0] - INFO[k][j]) < 5e-6 * max(1.0, abs(INFO[k][j])), key
                np.testing.assert_allclose(obs[0], OBS[k + 1], rtol=2e-6, atol=2e-5)
        print(f"config 1, {scheme} n_sub={n_sub} window {window or 'scheme'}: one-step state err {worst_x:.2e}, |d reward| {worst_r:.2e}, |d u| {worst_u:.2e}")
        assert worst_u < 1e-9 and worst_x < tol and worst_r < 20 * tol
        env.close()


def test_crop_noise_kernel_and_config5_step(golden, oracle):
    """BASELINE config 5: per-env +-10 % crop-parameter noise, re-drawn every step (noise.py).  (1) the device
    generator reproduces the Philox4x32-10 definition bit for bit; (2) a step with the drawn blocks equals the
    oracle step fed the same 208-vectors."""
    from test_controller_and_noise import expected_crop_noise
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    B = 96
    env = TomatoVecEnv(B, weather=w, dtype="float32", scheme="rk4", n_sub=256, season_length=1, uncertainty_scale=0.2, seed=4242,
                       auto_reset=False)
    env.reset()
    rng = np.random.default_rng(3)
    for k in range(2):
        acts = rng.uniform(-1, 1, (B, 6)).astype(np.float32)
        x_prev = env.x.double().cpu().numpy().copy(); u_prev = env.u.double().cpu().numpy().copy()
        env.step(acts)
        crop = env.crop_T[:, :B].cpu().numpy()
        exp = expected_crop_noise(env.p[128:162], B, 0.2, 4242, k)
        # identical Philox stream; the only freedom is fma contraction of p + noise*p on the device (<= 1 ulp)
        assert np.max(np.abs(crop - exp) / np.abs(exp)) < 1.3e-7
        assert np.all(np.abs(crop[:13] / env.p[128:141, None] - 1) <= 0.1 + 1e-6)      # +-scale/2
        xg = env.x.double().cpu().numpy()
        for b in range(0, B, 11):
            p = env.p.astype(np.float64).copy(); p[128:162] = crop[:, b]
            u = np.clip(u_prev[b] + acts[b] * np.float32(0.1), 0, 1)
            ref = oracle.rk_sc_guarded(x_prev[b], u, w[k], p, 900.0, 256, 4, 4)[0]
            assert scaled_err(xg[b], ref) < 5e-5
    assert len(np.unique(crop[1])) > B // 2                                            # envs really differ
    assert env.metrics()["n_ode_fail"] == 0
    env.close()
    # long run: +-10 % noise on laiMax / sla moves cLeafMax across cLeaf every few steps (harvest switch flips on);
    # with the exact harvest sub-flow no environment may fail and leaf mass must stay physical
    env = TomatoVecEnv(4096, weather=w, dtype="float32", n_sub=256, season_length=10, uncertainty_scale=0.2, seed=7,
                       auto_reset=False)
    env.reset()
    import torch
    g = torch.Generator(device=env.device); g.manual_seed(5)
    for k in range(200):
        env.step_tensor(torch.rand(4096, 6, generator=g, device=env.device) * 2 - 1, want_obs=False)
    assert env.metrics()["n_ode_fail"] == 0
    cleaf = env.x[:, 23]
    assert torch.isfinite(env.x).all() and float(cleaf.min()) > 5e4 and float(cleaf.max()) < 1.4e5
    env.close()


def test_config5_at_full_size_under_the_controlled_scheme(golden, oracle):
    """BASELINE config 5 at ITS size (B = 65 536, fp32, the shipped default scheme: the five-stage 2N scheme at n_sub 128 with stability control, guard and
    per-env crop blocks re-drawn every step): properties that do not depend on the size -- every drawn block within +-10 %
    (p144 derived), no failed integration over 40 steps, finite states, physical leaf mass -- and 24 environments picked across
    the batch checked for one step against the oracle's restatement of the SAME controlled scheme fed their 208-vectors."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    B = 65536
    env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=10, uncertainty_scale=0.2, seed=99, auto_reset=False)
    assert env.n_sub == 128 and env.scheme == "ls5" and env.window == 0
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(17)
    for k in range(39):
        env.step_tensor(torch.rand(B, 6, generator=gen, device=env.device) * 2 - 1, want_obs=False)
    pick = np.linspace(0, B - 1, 24).astype(int)
    x_prev = env.x[pick].double().cpu().numpy().copy(); u_prev = env.u[pick].double().cpu().numpy().copy()
    acts = torch.rand(B, 6, generator=gen, device=env.device) * 2 - 1
    env.step_tensor(acts, want_obs=False)
    crop = env.crop_T[:, :B].cpu().numpy()
    ratio = crop / env.p[128:162, None]
    assert np.all(np.abs(np.delete(ratio, 16, axis=0) - 1) <= 0.1 + 1e-6)
    np.testing.assert_allclose(crop[16], crop[13] / crop[14], rtol=3e-7)                  # p144 = p141 / p142 (noise.py:22)
    assert 0.04 < np.std(ratio[0]) < 0.07                                                # U(-0.1, 0.1): sigma 0.0577
    xg = env.x[pick].double().cpu().numpy(); a = acts[pick].cpu().numpy()
    worst = 0.0
    for j, b in enumerate(pick):
        p = env.p.astype(np.float64).copy(); p[128:162] = crop[:, b]
        u = np.clip(u_prev[j] + a[j] * np.float32(0.1), 0, 1)
        ref, retries, refined, failed = oracle.rk_sc_guarded(x_prev[j], u, w[39], p, 900.0, 128, 5, 2)
        assert not failed
        worst = max(worst, scaled_err(xg[j], ref))
    m = env.metrics()
    print(f"config 5 at B = 65 536: 40 steps, failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}, refined sub-steps "
          f"{m['n_refined_substeps']:.0f}; 24 envs vs the oracle's controlled scheme with their own blocks: {worst:.1e}")
    assert worst < 5e-5
    assert m["n_ode_fail"] == 0 and m["n_env_steps"] == 40 * B
    cleaf = env.x[:, 23]
    assert torch.isfinite(env.x).all() and float(cleaf.min()) > 5e4 and float(cleaf.max()) < 1.4e5
    env.close()


@pytest.mark.parametrize("scheme,order,win,preset", [("ls5", 5, 2, "throughput"), ("ls5", 5, 1, "parity"), ("rk4", 4, 4, "throughput"), ("rk2", 2, 4, "throughput"),
                                                     ("rk3", 3, 3, "throughput")])
def test_fp64_per_env_crop_blocks_in_every_scheme(golden, oracle, scheme, order, win, preset):
    """The fp64 kernels that take PER-ENVIRONMENT crop constants (a per-quad record in LDS; `step_kernel_quad<double, ..., CROP>` and
    `evalf_kernel_quad<double, ..., CROP>`, one instantiation per scheme) against the CPU checker's restatement fed each
    environment's own 208-vector: 48 environments with +-10 % crop noise, three env-steps from a spun-up state through glgym_step,
    and the same tuples through glgym_evalF with per-row parameter blocks, to rounding level."""
    import torch
    from gl_gym_amd import GreenLight
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    B = 48
    env = TomatoVecEnv(B, weather=w, dtype="float64", scheme=scheme, preset=preset, season_length=2, uncertainty_scale=0.2, seed=5, start_rows=[0, 130, 400],
                       auto_reset=False)
    env.reset()
    m = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, preset=preset)
    assert (m.n_sub, m.window) == (env.n_sub, env.window)
    m.set_verify("never")                                      # the action path below integrates guarded, unverified
    gen = torch.Generator(device=env.device); gen.manual_seed(23)
    w_off = env.w_off_t.cpu().numpy()
    worst_step, worst_evalf = 0.0, 0.0
    for k in range(3):
        x_prev = env.x.double().cpu().numpy().copy()
        env.step_tensor(torch.rand(B, 6, generator=gen, device=env.device) * 2 - 1, want_obs=False)
        crop = env.crop_T[:, :B].double().cpu().numpy()          # the blocks the kernel integrated with (re-drawn every step)
        u = env.u.double().cpu().numpy()
        xg = env.x.double().cpu().numpy()
        P = np.tile(env.p.astype(np.float64), (B, 1)); P[:, 128:162] = crop.T
        D = np.array([w[w_off[b] + k] for b in range(B)])
        Y = m.evalF_batch(x_prev, u, D, P)
        for b in range(B):
            ref, retries, refined, failed = oracle.rk_sc_guarded(x_prev[b], u[b], D[b], P[b], 900.0, env.n_sub, order, win)
            assert not failed
            worst_step = max(worst_step, scaled_err(xg[b][None], ref[None]))
            worst_evalf = max(worst_evalf, scaled_err(np.asarray(Y[b])[None], ref[None]))
    print(f"fp64 per-env crop blocks, {scheme} {preset}: glgym_step {worst_step:.1e}, glgym_evalF {worst_evalf:.1e} vs the checker's scheme")
    assert worst_step < 1e-10 and worst_evalf < 1e-10
    assert np.std(crop[0] / env.p[128]) > 0.03                   # the blocks really differ between environments
    env.close(); m.close()


def test_stability_control_in_storm(golden, oracle):
    """Wind 19.5 m/s with vents and screens open pushes the top-compartment exchange rate past RK4-256's stability
    limit during the step: the plain fixed-step scheme overflows there.  The stability control gives those windows
    smaller sub-steps (counted in n_refined_substeps), nothing is retried or flagged, and the result equals the oracle's
    restatement of the controlled scheme."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"].copy()
    w[:, 4] = 19.5; w[:, 1] = 2.0; w[:, 0] = 0.0
    B = 64
    env = TomatoVecEnv(B, weather=w, dtype="float32", scheme="rk4", n_sub=256, season_length=1, auto_reset=False)
    env.reset()
    p = env.p.astype(np.float64)
    ctrl = np.tile(np.array([0.9, 0.1, 0.0, 0.95, 0.5, 0.05]), (B, 1))
    plain_failed = False
    for k in range(12):
        x_prev = env.x.double().cpu().numpy().copy()
        env.step_raw_control(ctrl)
        ref, retries, refined, failed = oracle.rk_sc_guarded(x_prev[0], ctrl[0], w[k], p, 900.0, 256, 4, 4)
        plain_failed |= not np.all(np.isfinite(oracle.rk_lagged(x_prev[0], ctrl[0], w[k], p, 900.0, 256, 4, 4)))     # fixed step, no control
        assert np.all(np.isfinite(ref)) and not failed
        assert scaled_err(env.x[0].double().cpu().numpy(), ref) < 5e-5, k
    m = env.metrics()
    assert plain_failed, "the scenario no longer leaves RK4-256's stability region; pick a harsher one"
    assert m["n_ode_fail"] == 0 and m["n_refined_substeps"] >= B
    env.close()


def test_run_time_configuration_dt300_no_forecast(golden):
    """The reference's own timing harness (gl_gym/experiments/run_time.py:19-27) runs dt = 300 s, pred_horizon = 0,
    season 10 days, raw controls.  Same sequencing / reward scaling / observation layout (Np = 0 -> 23 floats) against the
    env oracle, with n_sub = 86 (h = 3.49 s)."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from oracle.gl_env_oracle import OracleTomatoEnv, INFO_KEYS
    w900 = golden("rollout_10day")["weather"]
    w = np.repeat(w900, 3, axis=0)[:3000]                  # a 300-s grid (zero-order hold of the 900-s fixture rows)
    B = 32
    env = TomatoVecEnv(B, weather=w, dtype="float64", dt=300.0, scheme="rk4", n_sub=86, season_length=0.02, pred_horizon=0,
                       start_rows=[0, 12], start_days=[0.0, 12 * 300 / 86400], seed=3, auto_reset=False)
    assert env.N == 5 and env.Np == 0 and env.obs_dim == 23
    obs0 = env.reset()
    w_off = env.w_off_t.cpu().numpy(); sd = env.start_day_t.cpu().numpy()
    assert abs(env.fixed_costs - (15 + 0.015 + 0.07 * 116 + 2) / 365 / (86400 // 300)) < 1e-15     # rewards.py:149-155
    rng = np.random.default_rng(2)
    orcs = []
    for b in range(0, B, 5):
        o = OracleTomatoEnv(weather=w[w_off[b]:], p=env.p, season_length=0.02, dt=300.0, pred_horizon=0,
                            integrator="rk4", n_sub=86, train_years=[0], train_days=[float(sd[b])], seed=0)
        ob = o.reset()
        assert ob.shape == (23,) and np.allclose(ob, obs0[b], rtol=2e-6, atol=1e-6)
        orcs.append((b, o))
    for k in range(6):
        ctrl = rng.uniform(0, 1, (B, 6))
        obs, rew, dones, info = env.step_raw_control(ctrl)
        for b, o in orcs:
            ob, r, term, inf = o.step_raw_control(ctrl[b])
            assert scaled_err(env.x[b].cpu().numpy(), o.x) < 1e-8
            assert abs(r - rew[b]) < 1e-6 and term == bool(dones[b])
            assert np.allclose(ob, obs[b], rtol=2e-6, atol=2e-5)
            for j, key in enumerate(INFO_KEYS):
                assert abs(inf[key] - info[j, b]) < 1e-6 * max(1.0, abs(inf[key])), key
    assert dones.all()
    env.close()


@pytest.mark.parametrize("pred_horizon,Np", [(1.0, 96), (0.26, 24), (0.0105, 1)])
def test_other_forecast_horizons_against_the_env_oracle(golden, pred_horizon, Np):
    """pred_horizon values other than the yml's 0.5 day and the harness's 0: the forecast block is 5 x Np floats (Np = int(pred_horizon
    x 86400 / dt), base_env.py:89), its rows come from weather[timestep + 1 ...] -- row width, LDS span and gather bounds of obs_kernel all depend
    on it.  Reset and three steps against the numpy env oracle for a few environments with their own episode starts."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from oracle.gl_env_oracle import OracleTomatoEnv
    w = golden("rollout_10day")["weather"]
    B = 24
    env = TomatoVecEnv(B, weather=w, dtype="float64", season_length=0.05, pred_horizon=pred_horizon, start_rows=[0, 7, 40],
                       start_days=[0.0, 7 * 900 / 86400, 40 * 900 / 86400], seed=5, auto_reset=False)
    assert env.Np == Np and env.obs_dim == 23 + 5 * Np
    obs0 = env.reset()
    w_off = env.w_off_t.cpu().numpy(); sd = env.start_day_t.cpu().numpy()
    assert len(set(w_off.tolist())) == 3
    rng = np.random.default_rng(4)
    orcs = []
    for b in range(0, B, 5):
        o = OracleTomatoEnv(weather=w[w_off[b]:], p=env.p, season_length=0.05, pred_horizon=pred_horizon, integrator="rk4", n_sub=256,
                            train_years=[0], train_days=[float(sd[b])], seed=0)
        ob = o.reset()
        assert ob.shape == (23 + 5 * Np,) and np.allclose(ob, obs0[b], rtol=2e-6, atol=1e-6)
        orcs.append((b, o))
    for k in range(3):
        ctrl = rng.uniform(0, 1, (B, 6))
        obs, rew, dones, info = env.step_raw_control(ctrl)
        for b, o in orcs:
            ob, r, term, inf = o.step_raw_control(ctrl[b])
            assert np.allclose(ob[23:], obs[b][23:], rtol=2e-6, atol=2e-5)          # the forecast block: exact up to float32
            assert np.allclose(ob[7:23], obs[b][7:23], rtol=2e-6, atol=2e-5)        # controls, weather, clocks
            assert scaled_err(env.x[b].cpu().numpy(), o.x) < 1e-4                   # (two different sub-steppers: the state only loosely)
    env.close()


def test_custom_reward_prices_and_constraints(golden):
    """Prices, dmfm and the constraint box come from configs/envs/TomatoEnv.yml in the reference; non-default values must
    flow through glgym_set_reward into the kernel epilogue exactly like rewards.py uses them."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from oracle.gl_env_oracle import OracleTomatoEnv, INFO_KEYS
    w = golden("rollout_10day")["weather"]
    rp = dict(elec_price=0.21, heating_price=0.05, co2_price=0.12, fruit_price=2.4, dmfm=0.0627)
    cs = dict(co2_min=500., co2_max=1000., temp_min=17., temp_max=20., rh_min=60., rh_max=75.)
    env = TomatoVecEnv(16, weather=w, dtype="float64", scheme="rk4", n_sub=256, season_length=1, reward_params=rp, constraints=cs,
                       auto_reset=False)
    env.reset()
    orc = OracleTomatoEnv(weather=w, p=env.p, season_length=1, integrator="rk4", n_sub=256, train_years=[0],
                          train_days=[0.0], seed=0, reward_params=rp, constraints=cs)
    orc.reset()
    # (the reference computes these from float32 parameter scalars: float32 under NumPy >= 2, float64 under its pinned 1.26)
    assert abs(env.max_profit / float(orc.reward.max_profit) - 1) < 2e-7
    assert abs(env.min_profit / float(orc.reward.min_profit) - 1) < 2e-7
    rng = np.random.default_rng(8)
    seen_violation = False
    for k in range(8):
        a = rng.uniform(-1, 1, 6).astype(np.float32)
        obs, rew, dones, infos = env.step(np.tile(a, (16, 1)))
        ob, r, term, info = orc.step(a)
        assert abs(r - rew[0]) < 1e-6
        for key in INFO_KEYS:
            assert abs(info[key] - infos[0][key]) < 1e-6 * max(1.0, abs(info[key])), key
        seen_violation |= info["temp_violation"] > 0 or info["rh_violation"] > 0
    assert seen_violation                  # the tightened box is actually violated, so the penalty path is exercised
    env.close()


@pytest.mark.gpu
def test_ode_pipe_variant_and_nd14_rows(golden, oracle):
    """SURVEY 8(f-4).  (1) GLGYM_ODE_PIPE against the reference-text vectors of ODE_pipe (ode.hpp:126-263) and the tight
    300 s step maps; (2) nd = 14 rows with the default ODE variant behave exactly like their first 10 columns (what the
    reference's compiled module does in experiments/gl_predefined_controls.py: it integrates ODE, the extra columns
    ride along); (3) the batched env on 14-column weather, step_raw_control_pipeinput, against the oracle's scheme."""
    import torch
    from gl_gym_amd import GreenLight
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("pipe_kat")
    X, U, D, P, DX, XT = g["X"], g["U"], g["D14"], g["P"], g["DX"], g["X_tight300"]
    n = len(XT)
    for dtype, tol_rhs, tol_step in (("float64", 1e-11, 1.3e-5), ("float32", 2e-4, 3e-5)):
        m = GreenLight(28, 6, 14, 208, 300.0, dtype=dtype, scheme="rk4", n_sub=256, variant="ode_pipe")
        for grp in (0, 1):                                    # even tuples: default p; odd: the MATLAB-comparison overrides
            idx = np.arange(grp, len(X), 2)
            if not np.array_equal(P[idx[0]], P[idx[-1]]):     # crop-noise tuples differ: one row at a time
                idx = idx[[np.array_equal(P[i], P[idx[-1]]) for i in idx]]
            m.set_params(P[idx[-1]])
            dx = m.rhs(X[idx], U[idx], D[idx])
            sc = np.maximum(np.abs(DX).max(axis=0), 1e-30)       # same scaling as test_rhs_matches_reference_text_vectors
            assert np.max(np.abs(dx - DX[idx]) / sc) < tol_rhs, (dtype, grp)
            assert np.all(dx[:, 19] == 0.0)
        got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(n)])
        err = scaled_err(got, XT)
        print(f"ODE_pipe {dtype}: one-step (300 s) err vs tight {err:.2e}")
        assert err < tol_step
        m.close()
    # (2) default variant, 14-column rows
    a = GreenLight(28, 6, 14, 208, 300.0, n_sub=256)             # (the default scheme, ls5)
    b = GreenLight(28, 6, 10, 208, 300.0, n_sub=256)
    np.testing.assert_array_equal(a.evalF_batch(X[:8], U[:8], D[:8]), b.evalF_batch(X[:8], U[:8], D[:8, :10]))
    with pytest.raises(Exception):
        GreenLight(28, 6, 10, 208, 300.0, scheme="rk4", variant="ode_pipe")          # needs the measured-pipe columns
    a.close(); b.close()
    # (2b) the variant with DEFAULT arguments (the default scheme "ls5" has no ODE_pipe kernels -- the constructors resolve
    # scheme=None to "rk4" for this variant, at their own preset counts, and refuse an explicit other scheme by name)
    m = GreenLight(28, 6, 14, 208, 300.0, variant="ode_pipe")
    assert (m.scheme, m.preset, m.n_sub) == ("rk4", "parity", 216)
    got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(n)])
    assert scaled_err(got, XT) < 3e-5            # rk4's parity count scaled to dt = 300 s (216), one unverified attempt: 2.1e-5 (at 256: < 1.3e-5, above)
    m.close()
    with pytest.raises(ValueError, match="ode_pipe"):
        GreenLight(28, 6, 14, 208, 300.0, variant="ode_pipe", scheme="ls5")
    # (3) env: 14-column weather table, controls held, pipe tracking on most rows
    w10 = golden("rollout_10day")["weather"][:200]
    rng = np.random.default_rng(5)
    w14 = np.concatenate([w10, np.column_stack([rng.uniform(35, 65, 200), rng.uniform(25, 45, 200),
                                                (np.arange(200) % 9 == 4) * 1.0, np.zeros(200)])], axis=1)
    w14[::13, 10] = 0.0
    p = P[1]
    for dtype in ("float64", "float32"):
        e0 = TomatoVecEnv(4, weather=w14, params=p, dt=300.0, season_length=0.05, pred_horizon=0.02, dtype=dtype, model_variant="ode_pipe",
                          auto_reset=False)                                      # default scheme / counts of the variant
        assert e0.scheme == "rk4" and e0.n_sub == (216 if dtype == "float64" else 80)
        e0.reset()
        xs0, _ = e0.step_raw_control_pipeinput(np.full((4, 6), 0.5))
        assert np.all(np.isfinite(xs0))
        e0.close()
    with pytest.raises(ValueError, match="ode_pipe"):
        TomatoVecEnv(4, weather=w14, params=p, dt=300.0, season_length=0.05, pred_horizon=0.02, model_variant="ode_pipe", scheme="ls5")
    env = TomatoVecEnv(4, weather=w14, params=p, dt=300.0, season_length=0.05, pred_horizon=0.02, dtype="float64",
                       scheme="rk4", n_sub=256, model_variant="ode_pipe", auto_reset=False)
    ref_env = TomatoVecEnv(4, weather=w10, params=p, dt=300.0, season_length=0.05, pred_horizon=0.02, dtype="float64",
                           scheme="rk4", n_sub=256, auto_reset=False)
    np.testing.assert_array_equal(env.reset(), ref_env.reset())           # obs / reset read the 14-wide rows correctly
    x = env.x[0].double().cpu().numpy()
    p64 = np.asarray(env.p, dtype=np.float64)
    for k in range(env.N + 1):
        u = rng.uniform(0, 1, 6)
        xs, term = env.step_raw_control_pipeinput(np.repeat(u[None], 4, 0))
        x = oracle.rk_sc_guarded(x, u, w14[k], p64, 300.0, 256, 4, 4, pipe=True)[0]
        assert scaled_err(xs[0], x) < 1e-9 and np.array_equal(xs[0], xs[3])
        assert bool(term[0]) == (k == env.N)
    env.close(); ref_env.close()



def test_rk2_scheme_matches_oracle_restatement(golden, oracle):
    """GLGYM_SCHEME_RK2 (explicit midpoint, tier 2b and harvest flow shared by four sub-steps) through glgym_evalF
    against the oracle's independent restatement of the same scheme, and against the tight one-step solutions."""
    from gl_gym_amd import GreenLight
    g = golden("step_tight")
    X, U, D, P, XT = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["X_tight"]
    for dtype, tol_o, tol_t in (("float64", 1e-9, 3e-5), ("float32", 3e-5, 4e-5)):
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme="rk2", preset="throughput")
        assert m.n_sub == 336                  # the scheme's nominal count (throughput preset)
        m.set_n_sub(360)
        got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(len(X))])
        ref = np.array([oracle.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, 360, 2, 4)[0] for i in range(len(X))])
        print(f"rk2 {dtype}: vs oracle scheme {scaled_err(got, ref):.2e}, vs tight {scaled_err(got, XT):.2e}")
        assert scaled_err(got, ref) < tol_o
        assert scaled_err(got, XT) < tol_t
        m.set_n_sub(357)                       # n_sub is rounded up to a multiple of the 4-sub-step window
        np.testing.assert_array_equal(np.array(m.evalF(X[0], U[0], D[0], P[0])), got[0])
        m.close()


def test_rk3_scheme_matches_oracle_restatement(golden, oracle):
    """GLGYM_SCHEME_RK3 (the exponential three-stage scheme, tier 2b and harvest flow shared by three sub-steps) through glgym_evalF against
    the oracle's independent restatement of the same scheme, and against the tight one-step solutions."""
    from gl_gym_amd import GreenLight
    g = golden("step_tight")
    X, U, D, P, XT = g["X"], g["U"], g["D"], g["P"].astype(np.float64), g["X_tight"]
    for dtype, tol_o, tol_t in (("float64", 1e-9, 3e-5), ("float32", 3e-5, 4e-5)):          # measured 2.7e-5 in fp64
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme="rk3", preset="throughput")
        assert m.n_sub == 270                  # the scheme's nominal count (throughput preset)
        got = np.array([m.evalF(X[i], U[i], D[i], P[i]) for i in range(len(X))])
        ref = np.array([oracle.rk_sc_guarded(X[i], U[i], D[i], P[i], 900.0, 270, 3, 3)[0] for i in range(len(X))])
        print(f"rk3 {dtype}: vs oracle scheme {scaled_err(got, ref):.2e}, vs tight {scaled_err(got, XT):.2e}")
        assert scaled_err(got, ref) < tol_o
        assert scaled_err(got, XT) < tol_t
        m.set_n_sub(268)                       # n_sub is rounded up to a multiple of the 3-sub-step window
        np.testing.assert_array_equal(np.array(m.evalF(X[0], U[0], D[0], P[0])), got[0])
        m.close()
    assert GreenLight(28, 6, 10, 208, 300.0, scheme="rk3", preset="throughput").n_sub == 90 and GreenLight(28, 6, 10, 208, 900.0, scheme="rk2", preset="throughput").n_sub == 336


@pytest.mark.parametrize("scheme,order,win,preset", [("ls5", 5, 2, "throughput"), ("ls5", 5, 1, "parity"), ("rk4", 4, 4, "throughput"), ("rk2", 2, 4, "throughput"),
                                                     ("rk3", 3, 3, "throughput")])
def test_fp64_step_kernel_tracks_oracle_scheme_step_by_step(golden, oracle, scheme, order, win, preset):
    """The fp64 kernels sit at the register limit and hipcc 7.2 has miscompiled them before (round 4: builds
    with the default parameter block compiled in computed wrong slow states -- traced to the max-ilp scheduler flag, since removed): every env-step of a short
    rollout must agree with the oracle's restatement of the same scheme to rounding level, through glgym_step AND glgym_evalF (two
    different kernels around the same four-lanes-per-environment integrator)."""
    import torch
    from gl_gym_amd import GreenLight
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("rollout_3day_synth")
    A, W, XR = g["actions"], g["weather"], g["X"]
    p = golden("params_default")["p"].astype(np.float64)
    env = TomatoVecEnv(64, weather=W, dtype="float64", scheme=scheme, preset=preset, season_length=3, auto_reset=False)
    env.reset()
    m = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, preset=preset)
    assert (m.n_sub, m.window) == (env.n_sub, env.window) and (env.window or win) == win
    u = np.zeros(6)
    for k in range(16):
        x_prev = env.x[0].double().cpu().numpy().copy()
        env.step_tensor(torch.as_tensor(np.repeat(A[k][None], 64, 0), device=env.device), want_obs=False)
        u = np.clip(u + A[k] * np.float32(0.1), 0, 1)
        ref = oracle.rk_sc_guarded(x_prev, u, W[k], p, 900.0, env.n_sub, order, win)[0]
        sc = np.maximum(np.abs(ref), 1e-3 * np.abs(XR).max(axis=0))
        assert np.max(np.abs(env.x[0].double().cpu().numpy() - ref) / sc) < 1e-11, (scheme, k, "glgym_step")
        assert np.max(np.abs(np.array(m.evalF(x_prev, u, W[k], p)) - ref) / sc) < 1e-11, (scheme, k, "glgym_evalF")
    env.close(); m.close()


def test_default_n_sub_scales_with_dt(golden, oracle):
    """Without an explicit n_sub the nominal sub-step keeps its length for any dt (throughput preset of the default scheme: 7.03 s -- 44
    sub-steps at the dt = 300 s of the reference's experiments/run_time.py, 256 at 1 800 s; the parity preset GreenLight() defaults
    to: 4.7 s -- 64 / 384); accuracy against plain RK4 with 8 192 sub-steps."""
    from gl_gym_amd import GreenLight
    g = golden("step_tight")
    X, U, D, P = g["X"], g["U"], g["D"], g["P"].astype(np.float64)
    scale = 1e-3 * np.abs(X).max(axis=0)
    for dt, preset, n_expect in ((300.0, "throughput", 44), (1800.0, "throughput", 256), (300.0, "parity", 64), (1800.0, "parity", 384)):
        for dtype in ("float64", "float32"):
            m = GreenLight(28, 6, 10, 208, dt, dtype=dtype, preset=preset)
            assert m.n_sub == n_expect and m.scheme == "ls5"
            for i in (0, 7, 19, 33):
                got = np.array(m.evalF(X[i], U[i], D[i], P[i]))
                ref = oracle.rk4_split(X[i], U[i], D[i], P[i], dt, 8192)
                assert float(np.max(np.abs(got - ref) / np.maximum(np.abs(ref), scale))) < 5e-5, (dt, dtype, i)
            m.close()


@pytest.mark.parametrize("scheme", ["ls5", "rk4", "rk2", "rk3"])
def test_two_waves_per_simd_build_matches_one_wave_build(scheme):
    """glgym_set_occupancy(h, 2) makes float32 default-parameter launches take the `__launch_bounds__(64, 2)` build of
    step_kernel (256 registers, the rest spilled to scratch).  Same source, other register allocation: its results must
    equal those of the one-wave build environment by environment, over several steps with observations, auto-reset bookkeeping
    and metrics.  Both batches are beyond the quad kernel's range (B > 16 384) so that both run one lane per environment; the
    large one = several copies of the small one."""
    import os
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    w = synthetic_weather(n_rows=4000)
    starts = list(range(0, 2000, 37))
    Bs, Bl = 20480, 81920
    small = TomatoVecEnv(Bs, weather=w, dtype="float32", scheme=scheme, season_length=1, start_rows=starts, seed=5,
                         auto_reset=False)
    large = TomatoVecEnv(Bl, weather=w, dtype="float32", scheme=scheme, season_length=1, start_rows=starts, seed=5,
                         auto_reset=False)
    large.set_occupancy(2)                   # handle state (round 5; GLGYM_OCC is only its initial value, read at glgym_create)
    small.reset(); large.reset()
    # the large batch = 64 copies of the small one (same start rows, states and actions per copy)
    rep = Bl // Bs
    large.w_off_t.copy_(small.w_off_t.repeat(rep)); large.start_day_t.copy_(small.start_day_t.repeat(rep))
    large.x_T[:, :Bl] = small.x_T[:, :Bs].repeat(1, rep); large.u_T[:, :Bl] = small.u_T[:, :Bs].repeat(1, rep)
    g = torch.Generator(device=small.device); g.manual_seed(3)
    worst = 0.0
    for k in range(4):
        a = torch.rand(Bs, 6, generator=g, device=small.device) * 2 - 1
        o_s, r_s, d_s, i_s = small.step_tensor(a)
        o_l, r_l, d_l, i_l = large.step_tensor(a.repeat(rep, 1))
        # column-scaled differences (conftest.scaled_err's metric): rounding-level, the two builds order a few float32
        # operations differently; measured 2e-7 ... 1e-6, against 1e-5 ... 2e-5 of either build to the float64 kernel
        # (reward and the profit entries of info are differences that pass through zero: scaled by the column maximum)
        for name, got, ref, env_major, floor in (("x", large.x_T[:, :Bl], small.x_T[:, :Bs].repeat(1, rep), False, 1e-3),
                                                 ("u", large.u_T[:, :Bl], small.u_T[:, :Bs].repeat(1, rep), False, 1e-3),
                                                 ("reward", r_l[None, :], r_s.repeat(rep)[None, :], False, 1.0),
                                                 ("info", i_l[:, :Bl], i_s[:, :Bs].repeat(1, rep), False, 1.0),
                                                 ("obs", o_l, o_s.repeat(rep, 1), True, 1e-3)):
            assert torch.isfinite(got).all()
            dim = 0 if env_major else 1
            scale = torch.maximum(ref.abs(), floor * ref.abs().amax(dim=dim, keepdim=True)).clamp_min(1e-30)
            err = float(((got - ref).abs() / scale).max())
            worst = max(worst, err)
            assert err < (1e-5 if name in ("reward", "info") else 5e-6), (name, k, err)      # (reward / profit: differences through zero)
        assert torch.equal(large.x_T[:, :Bs], large.x_T[:, Bl - Bs:Bl])            # copies agree bit for bit
        assert torch.equal(d_l, d_s.repeat(rep))
    # ... and through the verified ladder (raw controls: every environment takes at least two attempts, each re-reading the window state
    # the two-wave build keeps in LDS)
    ctrl = torch.rand(Bs, 6, generator=g, device=small.device, dtype=small.tdtype)
    small.set_verify("auto"); large.set_verify("auto")                             # (this module's handles are created unverified)
    small.step_tensor(controls_t=ctrl); large.step_tensor(controls_t=ctrl.repeat(rep, 1))
    ref = small.x_T[:, :Bs].repeat(1, rep)
    scale = torch.maximum(ref.abs(), 1e-3 * ref.abs().amax(dim=1, keepdim=True)).clamp_min(1e-30)
    err = float(((large.x_T[:, :Bl] - ref).abs() / scale).max())
    assert err < 5e-6, ("raw control", err)
    ms, ml = small.metrics(), large.metrics()
    assert ml["n_guard_retries"] >= Bl                                             # verified: at least one extra attempt each
    assert ml["n_env_steps"] == rep * ms["n_env_steps"] and ml["n_ode_fail"] == 0
    assert abs(ml["sum_reward"] - rep * ms["sum_reward"]) < 1e-4 * abs(rep * ms["sum_reward"]) + 1.0
    print(f"occupancy-2 build vs occupancy-1 build ({scheme}): worst relative difference {worst:.1e}")
    small.close(); large.close()


@pytest.mark.parametrize("B,n_steps", [(65536, 48), (131072, 16)])
def test_bench_workload_at_full_batch_against_oracle_and_fine_truth(oracle, B, n_steps):
    """(B = 131 072: the smallest batch the default dispatch gives the TWO-WAVES-PER-SIMD build of the kernel, window state in LDS --
    round 5 -- checked the same way: against the checker, not against the other build.)
    The workload bench.py TIMES (BASELINE configs[2]: B = 65 536, fp32, synthetic weather year, per-env episode starts, jittered
    states, fresh U(-1, 1) actions every step, the default dispatch = the one-lane kernel), checked instead of timed: 64 environments
    sampled across the wavefronts (one per 16 waves, rotating lane) are compared EVERY step with the CPU checker's restatement of the
    controlled scheme started from the kernel's own previous state (one-step maps: fp32 rounding), and 16 of them free-running over
    all 48 steps with plain classical RK4 at 8 192 sub-steps (the fine truth: the accuracy bar)."""
    from concurrent.futures import ThreadPoolExecutor
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
    starts = np.arange(0, 35040 - 5760 - 60, 96)
    env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=60, pred_horizon=0.5, seed=666, start_rows=starts, auto_reset=True)
    assert env.n_sub == 128 and env.scheme == "ls5" and env.window == 0      # bench.py's `value` configuration
    env.reset_tensor()
    dev = env.device
    env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1234)).to(env.tdtype))
    gen = torch.Generator(device=dev).manual_seed(666)
    pick = np.array([(B // 64) * i + (7 * i) % 64 for i in range(64)])               # 64 envs spread over the wavefronts, every lane residue
    p = env.p.astype(np.float64)
    w_off = env.w_off_t.cpu().numpy()[pick]
    x_true = env.x[pick].double().cpu().numpy().copy()                                # fine truth, free-running (first 16)
    pool = ThreadPoolExecutor(8)
    worst_one, worst_flags = 0.0, 0
    for k in range(n_steps):
        x_prev = env.x[pick].double().cpu().numpy().copy()
        env.action_t.uniform_(-1.0, 1.0, generator=gen)          # as bench.py's timed loop draws them
        env._launch_step(raw_control=False)
        u = env.u[pick].double().cpu().numpy()                                         # the control the kernel applied
        x_gpu = env.x[pick].double().cpu().numpy()
        flags = env.step_flags_t.cpu().numpy()[pick]
        ref = list(pool.map(lambda j: oracle.rk_sc_guarded(x_prev[j], u[j], w[w_off[j] + k], p, 900.0, 128, 5, 2, want_flags=True), range(64)))
        for j in range(64):
            assert not ref[j][3] and not (flags[j] & 128)
            worst_one = max(worst_one, scaled_err(x_gpu[j][None], ref[j][0][None]))
            worst_flags += int((flags[j] & 0xffff) != (ref[j][4] & 0xffff))          # same guard decisions (fp32 may flip one rarely)
        x_true[:16] = np.array(list(pool.map(lambda j: oracle.rk4(x_true[j], u[j], w[w_off[j] + k], p, 900.0, 8192), range(16))))
        env._launch_reset(env.done_t)
    e_true = scaled_err(env.x[pick[:16]].double().cpu().numpy(), x_true[:16])
    m = env.metrics()
    print(f"bench workload, B = {B} x {n_steps} steps: 64 sampled envs vs the oracle's scheme per step {worst_one:.1e} (guard words differing: "
          f"{worst_flags}); 16 envs free-running vs RK4-8192 truth {e_true:.1e}; failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}")
    assert worst_one < 5e-5 and worst_flags <= 2            # one-step maps in fp32 vs the fp64 restatement: 2.3e-5 measured
    assert e_true < 1e-4
    assert m["n_ode_fail"] == 0 and m["n_env_steps"] == B * n_steps
    env.close()
