"""GPU tests of the env-level drop-in surface: SB3 VecEnv calling convention (auto-reset, terminal_observation,
get_attr/env_method) and the single-env Gymnasium-style wrapper, following the reference's own tests/env_test.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vecenv_autoreset_and_terminal_observation(golden):
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd import INFO_KEYS
    w = golden("rollout_10day")["weather"]
    B = 8
    env = TomatoVecEnv(B, weather=w, dtype="float32", n_sub=224, season_length=0.05, start_rows=[0, 96], seed=1)
    assert env.N == 4 and env.num_envs == B and env.observation_space.shape == (263,) and env.action_space.shape == (6,)
    obs = env.reset()
    assert obs.shape == (B, 263) and obs.dtype == np.float32
    reset_obs = obs.copy()
    rng = np.random.default_rng(0)
    for k in range(5):                                   # episode = N + 1 = 5 steps (tests/env_test.py:77-92)
        obs, rew, dones, infos = env.step(rng.uniform(-1, 1, (B, 6)).astype(np.float32))
        assert rew.shape == (B,) and dones.shape == (B,) and len(infos) == B
        assert all(k_ in infos[0] for k_ in INFO_KEYS) and "controls" in infos[0]
        assert dones.all() == (k == 4)
    # SB3 semantics: the returned obs of a finished env is the first obs of the NEXT episode, the last obs of the old
    # one travels in info["terminal_observation"]
    for b in range(B):
        assert "terminal_observation" in infos[b]
        assert infos[b]["terminal_observation"][18] == 4.0          # "timestep" feature of the terminal step
        assert obs[b][18] == 0.0 and np.allclose(obs[b][7:13], 0.0)   # fresh episode: timestep 0, controls 0
        assert np.allclose(obs[b][4:7], reset_obs[b][4:7])          # crop state back at init_state
    assert env.get_attr("timestep") == [0] * B and env.get_attr("N", [0, 1]) == [4, 4]
    assert env.env_is_wrapped(object) == [False] * B and len(env.get_obs_names()) == 263
    assert len(env.get_attr("u")[0]) == 6 and len(env.get_attr("x", 0)[0]) == 28
    m = env.metrics()
    assert m["n_done"] == B and m["n_env_steps"] == 5 * B
    env.close()
    # large batches hand out a lazy list-like of the same dicts
    env2 = TomatoVecEnv(B, weather=w, dtype="float32", n_sub=224, season_length=0.05, start_rows=[0, 96], seed=1,
                        lazy_infos=True)
    env2.reset()
    rng = np.random.default_rng(0)
    for k in range(5):
        obs2, rew2, dones2, infos2 = env2.step(rng.uniform(-1, 1, (B, 6)).astype(np.float32))
    assert len(infos2) == B and np.array_equal(obs2, obs) and np.array_equal(rew2, rew)
    for b in (0, B - 1):
        assert set(infos2[b]) == set(infos[b]) and infos2[b]["EPI"] == infos[b]["EPI"]
        assert np.array_equal(infos2[b]["terminal_observation"], infos[b]["terminal_observation"])
    assert len(infos2[:3]) == 3 and sum(1 for _ in infos2) == B
    env2.close()


def test_single_env_wrapper_follows_reference_unit_tests(golden):
    """tests/env_test.py of the reference, on the B = 1 wrapper: max_profit known answer, reset/step shapes,
    action = -1 -> zero variable costs, action scaling inside bounds, episode length N + 1."""
    from gl_gym_amd.tomato_env import TomatoEnv
    w = golden("rollout_10day")["weather"]
    env = TomatoEnv(weather=w, dtype="float64", season_length=0.1, start_day=0.0)
    obs, info = env.reset(seed=42)
    assert len(obs) == env.observation_space.shape[0] and info == {} and env.timestep == 0 and not env.terminated
    assert abs(env.vec.max_profit - 0.328 * 900 * 1e-6 / 0.065 * 1.6) < 1e-7           # env_test.py:20-21
    obs, reward, terminated, truncated, info = env.step(np.ones(6, np.float32) * -1)
    assert isinstance(reward, float) and env.timestep == 1 and truncated is False      # env_test.py:44-57
    assert info["variable_costs"] == 0 and np.all(info["controls"] == 0)               # env_test.py:59-65
    a = env.action_space.sample()
    u = env.action_to_control(a)
    assert np.all(u >= env.u_min) and np.all(u <= env.u_max)                            # env_test.py:68-75
    steps, terminated = 1, False
    while not terminated and steps < 100:
        _, _, terminated, _, _ = env.step(env.action_space.sample())
        steps += 1
    assert terminated and steps == int(0.1 * 86400 // 900) + 1                          # env_test.py:77-92
    env.set_crop_state(cBuf=0, cLeaf=0.9e5, cStem=2.5e5, cFruit=2.8e5, tCanSum=3000)
    assert env.x[25] == 2.8e5
    env.close()


def test_scripts_that_edit_a_constructed_env_weather_parameters_state(golden):
    """experiments/run_time.py:36-48 and gl_predefined_controls.py:110-126 assign `env.weather_data`, `env.p` and `env.x` on an env that already
    exists.  On the B = 1 wrapper each assignment reaches the device (none is a silent no-op) with the reference's meaning: the new table is read
    from its row 0 at the current timestep, the new block drives the model, the state is replaced -- the next raw-control step equals, bit for
    bit, that of an env CONSTRUCTED with that table and block and put into that state."""
    from gl_gym_amd.tomato_env import TomatoEnv
    g = golden("holdout_runtime_dt300")
    w_a, w_b = golden("rollout_10day")["weather"], g["weather"]
    u = np.array([0.3, 0.1, 0.6, 0.2, 1.0, 0.4])
    env = TomatoEnv(weather=w_a, dtype="float64", season_length=1, start_day=0.0)
    env.reset(seed=1)
    env.weather_data = w_b
    env.p = g["p"]
    env.x = g["x0"]
    assert np.array_equal(env.weather_data, w_b) and np.array_equal(env.p, g["p"].astype(np.float32)) and np.array_equal(env.x, g["x0"])
    o1, r1, _, _, i1 = env.step_raw_control(u)
    ref = TomatoEnv(weather=w_b, params=g["p"], dtype="float64", season_length=1, start_day=0.0)
    ref.reset(seed=1)
    ref.x = g["x0"]
    o2, r2, _, _, i2 = ref.step_raw_control(u)
    assert np.array_equal(env.x, ref.x) and np.array_equal(o1, o2)                     # state and observation: bit for bit
    assert abs(r1 - r2) > 1e-3                                                         # the reward's SCALE stays the construction-time block's (rewards.py:82-83)
    assert i1["heat_cost"] == i2["heat_cost"] and i1["co2_cost"] == i2["co2_cost"]     # ... its per-step costs follow the new one (rewards.py:164-166)
    base = TomatoEnv(weather=w_a, dtype="float64", season_length=1, start_day=0.0)     # and none of it was a no-op: the untouched env goes elsewhere
    base.reset(seed=1)
    base.step_raw_control(u)
    assert np.abs(base.x - env.x).max() > 1.0
    with pytest.raises(ValueError):
        env.weather_data = w_b[:, :7]                                                  # the column count is the handle's nd
    with pytest.raises(ValueError):
        env.weather_data = w_b[:20]                                                    # shorter than an episode
    for e in (env, ref, base):
        e.close()


def test_greenlight_drop_in_signature(golden):
    """gl_gym.environments.models.greenlight_model.GreenLight contract (greenlight_model.cpp:130-136)."""
    from gl_gym_amd import GreenLight, GlgymError
    g = golden("params_default")
    m = GreenLight(28, 6, 10, 208, 900.0)
    x1 = m.evalF(list(g["x0"]), [0.0] * 6, list(g["d0"]), g["p"])       # lists / float32 array like the reference
    assert isinstance(x1, list) and len(x1) == 28
    x2 = m.evalF(x1, np.zeros(6), g["d0"], g["p"].astype(np.float64))   # feed the returned list back in
    assert np.all(np.isfinite(x2)) and abs(x2[27] - 2 * 900 / 86400) < 1e-12
    with pytest.raises(GlgymError):
        GreenLight(27, 6, 10, 208, 900.0)
    m.close()


def test_vecnormalize_on_device_matches_sb3_algorithm(golden):
    """VecNormalizeGPU (glgym_vecnorm kernels) vs the numpy restatement of SB3 2.6.0's VecNormalize, driven by the same
    raw observations / rewards / dones for 12 steps incl. an episode boundary (returns reset)."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.vec_normalize import VecNormalizeGPU
    from oracle.vecnorm_oracle import VecNormalizeOracle
    w = golden("rollout_10day")["weather"]
    B = 512
    env = TomatoVecEnv(B, weather=w, dtype="float32", n_sub=224, season_length=0.08, start_rows=[0, 30, 200], seed=9)
    vn = VecNormalizeGPU(env, clip_obs=10.0, gamma=0.9631)                 # gamma of configs/agents/ppo.yml:8
    orc = VecNormalizeOracle(B, env.obs_dim, clip_obs=10.0, gamma=0.9631)
    obs_n = vn.reset()
    ref_n = orc.reset(vn.get_original_obs().astype(np.float64))
    np.testing.assert_allclose(obs_n, ref_n, rtol=1e-5, atol=2e-5)
    rng = np.random.default_rng(0)
    saw_done = False
    for k in range(12):
        obs_n, rew_n, dones, infos = vn.step(rng.uniform(-1, 1, (B, 6)).astype(np.float32))
        raw_r = vn.get_original_reward().astype(np.float64)
        # with auto-reset the raw obs of finished envs is the next episode's first obs, exactly what SB3 would see
        ref_o, ref_r = orc.step(vn.get_original_obs().astype(np.float64), raw_r, dones)
        np.testing.assert_allclose(obs_n, ref_o, rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(rew_n, ref_r, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(vn.obs_rms.mean, orc.obs_rms.mean, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(vn.obs_rms.var, orc.obs_rms.var, rtol=1e-7, atol=1e-9)
        assert abs(vn.obs_rms.count - orc.obs_rms.count) < 1e-6
        assert abs(vn.ret_rms.var - orc.ret_rms.var) < 1e-9 * max(1.0, orc.ret_rms.var)
        np.testing.assert_allclose(vn.returns.cpu().numpy(), orc.returns, rtol=1e-9, atol=1e-12)
        saw_done |= bool(dones.any())
    assert saw_done and np.abs(obs_n).max() <= 10.0
    x = rng.standard_normal((4, env.obs_dim))
    np.testing.assert_allclose(vn.unnormalize_obs(x), orc.unnormalize_obs(x), rtol=1e-9)
    # eval-mode wrapper (training=False, norm_reward=False) leaves the statistics untouched
    vn.training, vn.norm_reward = False, False
    m0 = vn.obs_rms.mean.copy()
    _, rew_e, _, _ = vn.step(np.zeros((B, 6), np.float32))
    assert np.array_equal(vn.obs_rms.mean, m0) and np.allclose(rew_e, vn.get_original_reward())
    # save / VecNormalize.load round trip (experiments/evaluate_rl.py:31, callbacks' best_vecnormalize.pkl)
    import os, tempfile
    path = os.path.join(tempfile.mkdtemp(), "best_vecnormalize.pkl")
    vn.save(path)
    vn2 = VecNormalizeGPU.load(path, env)
    vn2.training, vn2.norm_reward = False, False
    assert np.array_equal(vn2.obs_rms.mean, vn.obs_rms.mean) and np.array_equal(vn2.obs_rms.var, vn.obs_rms.var)
    assert vn2.clip_obs == 10.0 and vn2.gamma == 0.9631
    raw = vn.get_original_obs()
    np.testing.assert_array_equal(vn2.normalize_obs(raw), vn.normalize_obs(raw))
    env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-12), ("float32", 2e-5)])
def test_rule_based_kernel_against_reference_vectors(golden, dtype, tol):
    """glgym_rule_based against controller_kat.npz = outputs of the reference's RuleBasedController.predict on 128
    random (x, d, hour_of_day, day_of_year) tuples.  The weather table is the fixture's d rows, one env per row, explicit
    clocks.  fp32 handles read x and d rounded to float (the rules themselves are evaluated in fp64)."""
    import ctypes as C
    import torch
    from gl_gym_amd import _lib as L
    from gl_gym_amd.baseline import RuleBasedController
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("controller_kat")
    n = len(g["X"])
    env = TomatoVecEnv(n, weather=np.vstack([g["D"], g["D"][-1:]]), dtype=dtype, season_length=1e-9, pred_horizon=0,
                       auto_reset=False)
    env.reset()
    env.x_T[:, :n] = torch.as_tensor(g["X"].T, device=env.device)
    env.w_off_t.copy_(torch.arange(n, dtype=torch.int32))
    hour = torch.as_tensor(g["hour"], dtype=torch.float64, device=env.device)
    doy = torch.as_tensor(g["doy"], dtype=torch.float64, device=env.device)
    env._launch_rule_based(RuleBasedController(), hour, doy)
    u = env.ctrl_T[:, :n].t().double().cpu().numpy()
    assert np.abs(u - g["U"]).max() < tol
    # derived clocks: timestep 7 of an episode started on day 59 -> hour 1.75, doy 59 + 7/96
    env.timestep_t.fill_(7)
    env.start_day_t.fill_(59.0)
    env.w_off_t.sub_(7).clamp_(min=0)
    ref = RuleBasedController().predict(env.x.double().cpu().numpy(), env.current_weather().double().cpu().numpy(),
                                        np.full(n, 7 * 0.25), np.full(n, 59.0 + 7 / 96.0))
    out_tol = 1e-12 if dtype == "float64" else 1e-7       # fp32 handles store the controls as float
    assert np.abs(env.rule_based_controls(RuleBasedController()).double().cpu().numpy() - ref).max() < out_tol
    # a zero proportional band is refused, not divided by
    with pytest.raises(L.GlgymError):
        env._launch_rule_based(RuleBasedController(co2Band=0))
    env.close()


@pytest.mark.gpu
def test_step_rule_based_closed_loop_matches_host_mirror(golden):
    """step_tensor(controller=...) == rule_based_controls + step_raw_control, and the day runs to its 97th step."""
    from gl_gym_amd.baseline import RuleBasedController
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    ctrl = RuleBasedController()
    a = TomatoVecEnv(16, weather=w, dtype="float64", season_length=1, auto_reset=False)
    b = TomatoVecEnv(16, weather=w, dtype="float64", season_length=1, auto_reset=False)
    a.reset(); b.reset()
    for k in range(97):
        b.x_T.copy_(a.x_T); b.u_T.copy_(a.u_T)       # teacher-forced: the closed loop amplifies ulp differences
        u_host = ctrl.predict(b.x.double().cpu().numpy(), b.current_weather().double().cpu().numpy(),
                              b.hour_of_day().cpu().numpy(), b.day_of_year().cpu().numpy())
        oa, ra, da, ia = a.step_rule_based(ctrl)
        ob, rb, db, _ = b.step_raw_control(u_host)
        assert np.abs(a.u.cpu().numpy() - u_host).max() < 1e-10      # device exp vs numpy exp through the steep bands
        np.testing.assert_allclose(oa, ob, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(ra, rb, rtol=0, atol=1e-8)
        assert bool(da[0]) == (k == 96)
    assert set(ia[0]) >= {"EPI", "controls"}
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_step_graph_replay_equals_eager_steps(golden, dtype):
    """capture_step_graph(): the five launches of a step replayed from one HIP graph give bit-identical tensors to
    step_tensor, across an episode boundary (auto-reset inside the graph)."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    kw = dict(weather=w, dtype=dtype, season_length=0.05, start_rows=[0, 40, 300], start_days=[0.0, 0.4167, 3.125], seed=3)
    a, b = TomatoVecEnv(200, **kw), TomatoVecEnv(200, **kw)      # float64: the kernels carry 73 KB of dynamic LDS
    a.reset_tensor(); b.reset_tensor()
    replay = b.capture_step_graph()
    assert torch.equal(a.x_T, b.x_T) and torch.equal(a.timestep_t, b.timestep_t)     # capture left the state alone
    g = torch.Generator(device=a.device); g.manual_seed(0)
    for k in range(12):                                   # N = 4: episodes end at steps 5 and 10
        act = torch.rand(200, 6, generator=g, device=a.device) * 2 - 1
        oa, ra, da, ia = a.step_tensor(act)
        ob, rb, db, ib = replay(act)
        for ta, tb in ((oa, ob), (ra, rb), (da, db), (ia, ib), (a.x_T, b.x_T), (a.term_obs_t, b.term_obs_t),
                       (a.w_off_t, b.w_off_t)):
            assert torch.equal(ta, tb), k
        assert bool(da.all()) == (k % 5 == 4)
    ma, mb = a.metrics(), b.metrics()                     # float atomics: order-dependent in the last bits
    assert all(abs(ma[k] - mb[k]) <= 1e-5 * max(1.0, abs(ma[k])) for k in ma)
    a.close(); b.close()


@pytest.mark.gpu
def test_c_abi_rejects_bad_arguments_and_handles_ragged_batches(golden):
    """Edge cases at the boundary: empty batch, ld < B, missing / ambiguous control input and out-of-range Np are status
    codes (never a launch); batch sizes that are not a multiple of the 64-lane wave or of the 16-row observation span
    (B = 1, 63, 65, 1000) give the same per-env results as a large batch."""
    import ctypes as C
    import torch
    from gl_gym_amd import _lib as L
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"]
    env = TomatoVecEnv(128, weather=w, dtype="float32", season_length=1, auto_reset=False)
    env.reset_tensor()
    lib, h, st = env._lib, env._h, env._stream()

    def step_args(**over):
        kw = dict(B=env.B, ld=env.ld, x=env.x_T.data_ptr(), u=env.u_T.data_ptr(), action=env.action_t.data_ptr(), control=None,
                  weather=env.weather_t.data_ptr(), weather_rows=env.weather_rows, w_off=env.w_off_t.data_ptr(),
                  timestep=env.timestep_t.data_ptr(), crop_p=None, N=env.N, reward=env.reward_t.data_ptr(),
                  info=env.info_T.data_ptr(), done=env.done_t.data_ptr(), metrics=None, step_flags=None)
        kw.update(over)
        return L.make_step_args(*[kw[f[0]] for f in L.StepArgs._fields_[1:]])

    x_before = env.x_T.clone()
    for bad in (dict(B=0), dict(B=-3), dict(ld=64), dict(x=None), dict(action=None), dict(control=env.ctrl_T.data_ptr()),
                dict(weather_rows=0), dict(reward=None)):
        assert lib.glgym_step(h, C.byref(step_args(**bad)), st) == L.EINVAL, bad
        assert b"glgym_step" in lib.glgym_last_error()
    assert lib.glgym_step(None, C.byref(step_args()), st) == L.EINVAL
    # a caller built against an older header (a shorter struct: no step_flags) is refused before any pointer is read (ABI 5)
    old = step_args()
    old.struct_size = C.sizeof(L.StepArgs) - 8
    assert lib.glgym_step(h, C.byref(old), st) == L.EINVAL and b"struct_size" in lib.glgym_last_error()
    torch.cuda.synchronize()
    assert torch.equal(env.x_T, x_before)                        # nothing was launched
    oa = L.ObsArgs(env.B, env.ld, env.x_T.data_ptr(), env.u_T.data_ptr(), env.weather_t.data_ptr(), env.weather_rows,
                   env.w_off_t.data_ptr(), env.timestep_t.data_ptr(), env.start_day_t.data_ptr(), 129,
                   env.obs_t.data_ptr(), None, None)
    assert lib.glgym_obs(h, C.byref(oa), st) == L.EINVAL          # Np beyond the kernel's LDS span
    assert lib.glgym_set_n_sub(h, 0) == L.EINVAL and lib.glgym_set_scheme(h, 7) == L.EINVAL
    env.close()
    # ragged batch sizes: env b of every batch follows the same trajectory (same start row, same actions)
    ref = None
    for B in (1000, 1, 63, 65):
        e = TomatoVecEnv(B, weather=w, dtype="float32", season_length=1, auto_reset=False)
        e.reset_tensor()
        a = torch.linspace(-1, 1, 6, device=e.device).repeat(B, 1)
        for _ in range(3):
            obs, rew, done, info = e.step_tensor(a)
        got = (e.x[0].clone(), obs[0].clone(), rew[0].clone(), e.x[B - 1].clone(), obs[B - 1].clone())
        if ref is None:
            ref = got
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
        assert torch.equal(got[3], ref[0]) and torch.equal(got[4], ref[1])       # last lane / last row of the ragged tail
        e.close()


def test_gymnasium_vector_env_facade(golden):
    """gymnasium.vector.VectorEnv surface (north_star; reference base class gl_gym/environments/base_env.py:14):
    reset(seed=, options=) -> (obs, infos), five-tuple step, dict-of-arrays infos with masks, SAME_STEP autoreset with
    final_obs / final_info, and the same numbers as the SB3-style TomatoVecEnv on the same seed."""
    from gl_gym_amd.vector_env import TomatoVectorEnv
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd import INFO_KEYS
    w = golden("rollout_10day")["weather"]
    B = 8
    kw = dict(weather=w, dtype="float32", n_sub=224, season_length=0.05, start_rows=[0, 96], seed=1)
    venv = TomatoVectorEnv(B, **kw)
    ref = TomatoVecEnv(B, **kw)
    assert venv.num_envs == B and venv.single_observation_space.shape == (263,) and venv.single_action_space.shape == (6,)
    assert venv.observation_space.shape == (B, 263) and venv.action_space.shape == (B, 6)
    assert venv.metadata["autoreset_mode"] == "same_step"
    obs, infos = venv.reset(seed=5, options={"ignored": True})
    ref.reset_tensor(5)
    assert obs.shape == (B, 263) and obs.dtype == np.float32 and infos == {}
    assert np.array_equal(obs, ref.obs_t.cpu().numpy())
    rng = np.random.default_rng(0)
    for k in range(5):                                               # episode = N + 1 = 5 steps
        a = rng.uniform(-1, 1, (B, 6)).astype(np.float32)
        obs, rew, term, trunc, infos = venv.step(a)
        o2, r2, d2, i2 = ref.step(a)
        assert obs.shape == (B, 263) and rew.shape == term.shape == trunc.shape == (B,) and not trunc.any()
        assert np.array_equal(obs, o2) and np.allclose(rew, r2) and np.array_equal(term, d2)
        for key in INFO_KEYS:
            assert infos[key].shape == (B,) and infos["_" + key].all()
            assert np.allclose(infos[key], [i2[b][key] for b in range(B)])
        assert infos["controls"].shape == (B, 6)
        # the control applied in THIS step, also for envs that finished and were reset (tomato_env.py:221)
        assert np.allclose(infos["controls"], [i2[b]["controls"] for b in range(B)])
        assert (term.all() if k == 4 else not term.any())
        if k < 4:
            assert "final_obs" not in infos
            u_expect = infos["controls"].copy()
    assert np.abs(infos["controls"]).max() > 0                       # not the zeros the reset wrote
    assert infos["_final_obs"].all() and infos["_final_info"].all()
    for b in range(B):
        assert infos["final_obs"][b][18] == 4.0 and obs[b][18] == 0.0     # terminal "timestep" feature vs fresh episode
        assert np.array_equal(infos["final_obs"][b], i2[b]["terminal_observation"])
    assert np.allclose(infos["final_info"]["EPI"], infos["EPI"])
    assert venv.get_attr("N") == (4,) * B and len(venv.call("get_obs_names")[0]) == 263
    venv.close(); ref.close()


def test_wrappers_follow_the_step_async_step_wait_protocol(golden):
    """SB3 drives a VecEnv as step_async + step_wait (and the reference's own wrapper relies on it,
    vec_env_wrappers.py:13-17): the wrappers' own processing must run on that path too -- normalised observations,
    updated running statistics, infos['episode'] -- not the inner env's raw step."""
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.vec_monitor import VecMonitorGPU
    from gl_gym_amd.vec_normalize import VecNormalizeGPU
    w = golden("rollout_10day")["weather"]
    B = 16

    def stack():
        return VecNormalizeGPU(VecMonitorGPU(TomatoVecEnv(B, weather=w, dtype="float32", n_sub=224, season_length=0.05,
                                                          start_rows=[0, 96], seed=3)), clip_obs=10.0, gamma=0.99)
    a_env, b_env = stack(), stack()
    oa, ob = a_env.reset(), b_env.reset()
    assert np.array_equal(oa, ob)
    rng = np.random.default_rng(1)
    for k in range(6):
        act = rng.uniform(-1, 1, (B, 6)).astype(np.float32)
        ra = a_env.step(act)
        b_env.step_async(act)
        rb = b_env.step_wait()
        for x, y in zip(ra[:3], rb[:3]):
            assert np.array_equal(x, y)
        assert np.abs(ra[0]).max() <= 10.0 + 1e-6                    # clipped, normalised observations
        if ra[2].any():
            assert all("episode" in rb[3][b] and "terminal_observation" in rb[3][b] for b in np.nonzero(rb[2])[0])
            assert all(np.abs(rb[3][b]["controls"]).max() > 0 for b in np.nonzero(rb[2])[0])
    assert a_env.obs_rms.count == b_env.obs_rms.count > B
    assert np.allclose(a_env.obs_rms.mean, b_env.obs_rms.mean)
    # save / load round trip (SB3 attribute names, see tests/test_make_env.py for SB3-written files)
    import os
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "vn.pkl")
        a_env.save(path)
        c_env = VecNormalizeGPU.load(path, b_env.venv)
        assert np.array_equal(c_env.obs_rms.mean, a_env.obs_rms.mean) and c_env.ret_rms.var == a_env.ret_rms.var
        assert c_env.gamma == 0.99 and c_env.clip_obs == 10.0
    a_env.close(); b_env.close()


def test_weather_pipeline_on_the_device(golden, tmp_path):
    """glgym_weather (unit conversions, daily light sum, daylight flags, PCHIP resample on the device) against the host
    loader and against the table the REFERENCE's load_weather_data produced from the same raw rows
    (tests/golden/weather_bleiswijk2009.npz: small_raw -> small_out)."""
    from gl_gym_amd.weather_device import WeatherPipeline
    from gl_gym_amd.utils import weather_from_raw, load_weather_data
    g = golden("weather_bleiswijk2009")
    cols = [str(c) for c in g["small_raw_cols"]]
    raw = g["small_raw"]
    col = lambda name: raw[:, cols.index(name)]  # noqa: E731
    args = (col("time"), col("global radiation"), col("air temperature"), col("RH"), col("wind speed"),
            col("sky temperature"))
    host = weather_from_raw(*args, 900.0, 10)
    wp = WeatherPipeline(dtype="float64")
    dev = wp.from_raw(*args, 900.0).cpu().numpy()
    assert dev.shape == host.shape == g["small_out"].shape
    sc = np.maximum(np.abs(host).max(axis=0), 1e-30)
    assert np.max(np.abs(dev - host) / sc) < 1e-12
    assert np.max(np.abs(dev - g["small_out"]) / sc) < 1e-12            # the reference's own output
    assert np.array_equal(dev[:, 0] == 0.0, host[:, 0] == 0.0)          # the iGlob clean-up hits the same samples
    # fp32 table and a wider row (ODE_pipe layout): same values to float precision, extra columns zero
    wp32 = WeatherPipeline(dtype="float32", nd=14)
    d32 = wp32.from_raw(*args, 900.0).cpu().numpy()
    assert d32.shape == (host.shape[0], 14) and np.all(d32[:, 10:] == 0)
    assert np.max(np.abs(d32[:, :10] - host) / sc) < 1e-6
    # through the CSV front end, same signature as the reference's loader; then drive an env from the device table
    wdir = tmp_path / "w" / "Testville"
    wdir.mkdir(parents=True)
    two = np.concatenate([raw, raw]); two[:, cols.index("time")] = 300.0 * np.arange(len(two))
    with open(wdir / "GL2009.csv", "w") as f:
        f.write(",".join(cols) + "\n")
        for r in two:
            f.write(",".join(repr(float(v)) for v in r) + "\n")
    t_dev = wp.load_weather_data(str(tmp_path / "w"), "Testville", "GL", 2009, 0, 1, 1, 900.0)
    t_host = load_weather_data(str(tmp_path / "w"), "Testville", "GL", 2009, 0, 1, 1, 900.0, 10)
    assert np.max(np.abs(t_dev.cpu().numpy() - t_host) / np.maximum(np.abs(t_host).max(axis=0), 1e-30)) < 1e-12
    wp.close(); wp32.close()


def test_host_infos_never_report_an_earlier_steps_controls_and_weather_may_be_a_list(golden):
    """(advisor, round 2) After an SB3-style step_wait() a later step_tensor() + host_infos() must report the controls of THAT step,
    not the copy kept for the earlier one; and `weather` may be a nested list."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    w = golden("rollout_10day")["weather"][:400]
    env = TomatoVecEnv(8, weather=w.tolist(), dtype="float32", season_length=1, auto_reset=True)
    assert env.nd == 10
    env.reset()
    rng = np.random.default_rng(0)
    env.step_async(rng.uniform(-1, 1, (8, 6)).astype(np.float32))
    _, _, _, infos = env.step_wait()
    u_first = np.array([infos[b]["controls"] for b in range(8)])
    out = env.step_tensor(torch.as_tensor(rng.uniform(-1, 1, (8, 6)).astype(np.float32), device=env.device))
    dones, infos2 = env.host_infos(out[2], out[3])
    u_now = env.u.double().cpu().numpy()
    got = np.array([infos2[b]["controls"] for b in range(8)])
    np.testing.assert_allclose(got, u_now, rtol=0, atol=1e-7)
    assert np.abs(got - u_first).max() > 1e-3
    env.close()


def test_environment_variables_are_handle_state_read_once():
    """GLGYM_LAYOUT / GLGYM_OCC / GLGYM_VERIFY are initial values read at glgym_create; toggling os.environ afterwards has no
    effect on an existing handle -- the Python layer says so (once) instead of silently ignoring it."""
    import os
    import warnings
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    env = TomatoVecEnv(64, weather=synthetic_weather(n_rows=400), season_length=1, auto_reset=False)
    env.reset()
    a = np.zeros((64, 6), np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        env.step(a)                                         # nothing changed: no warning
    old = os.environ.get("GLGYM_LAYOUT")
    os.environ["GLGYM_LAYOUT"] = "one"
    try:
        x_before = env.x.clone()
        with pytest.warns(RuntimeWarning, match="read once"):
            env.step(a)
        with warnings.catch_warnings():
            warnings.simplefilter("error")
            env.step(a)                                     # said once
        assert not bool((env.x == x_before).all())          # ... and the step ran (on the layout the handle was created with)
    finally:
        if old is None:
            os.environ.pop("GLGYM_LAYOUT", None)
        else:
            os.environ["GLGYM_LAYOUT"] = old
    env.close()
