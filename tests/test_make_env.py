"""Config-driven construction (gl_gym_amd/make_env.py): the reference's load_env_params / make_vec_env call shapes."""
import shutil

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture()
def cfg_dir(tmp_path, golden):
    """A config directory + a weather CSV tree in the reference's layout (<dir>/<location>/<source><year>.csv), the CSV
    rows being the raw Bleiswijk sample the weather fixture holds."""
    g = golden("weather_bleiswijk2009")
    cols = [str(c) for c in g["small_raw_cols"]]
    raw = np.concatenate([g["small_raw"], g["small_raw"]])           # 2 x 577 rows at 300 s: room for start days 0 and 1
    raw[:, cols.index("time")] = 300.0 * np.arange(len(raw))
    wdir = tmp_path / "weather" / "Testville"
    wdir.mkdir(parents=True)
    with open(wdir / "GL2009.csv", "w") as f:
        f.write(",".join(cols) + "\n")
        for r in raw:
            f.write(",".join(repr(float(v)) for v in r) + "\n")
    yml = (GOLDEN / "TomatoEnvSmall.yml").read_text().replace("WEATHER_DIR", str(tmp_path / "weather"))
    (tmp_path / "TomatoEnv.yml").write_text(yml)
    return tmp_path


def test_load_env_params_and_season_table(cfg_dir):
    from gl_gym_amd.make_env import load_env_params, season_table, _check_supported, OBSERVATION_MODULES
    from gl_gym_amd.utils import load_weather_data
    base, spec = load_env_params("TomatoEnv", str(cfg_dir))
    assert base["dt"] == 900 and base["nd"] == 10 and spec["reward_function"] == "GreenhouseReward"
    assert spec["observation_modules"] == OBSERVATION_MODULES
    assert load_env_params("GreenLightEnv", str(cfg_dir)) if (cfg_dir / "GreenLightEnv.yml").exists() else True
    table, rows, days, years = season_table(base["weather_data_dir"], "Testville", "GL", [2009], [0, 1], 0.5, 0.01, 900, 10)
    one = load_weather_data(base["weather_data_dir"], "Testville", "GL", 2009, 1, 0.5, 1, 900, 10)   # reset()'s call, day 1
    assert rows == [0, len(table) - len(one)] and days == [0.0, 1.0] and years == [2009, 2009]
    np.testing.assert_array_equal(table[rows[1]:], one)
    # unsupported configurations are refused, not approximated
    with pytest.raises(NotImplementedError):
        _check_supported("SomeOtherReward", OBSERVATION_MODULES, base)
    _check_supported("GreenhouseReward", OBSERVATION_MODULES[:-1], base)           # subsets / other orders are laid out
    _check_supported("GreenhouseReward", OBSERVATION_MODULES[:1] + OBSERVATION_MODULES[:0:-1], base)
    with pytest.raises(KeyError):
        _check_supported("GreenhouseReward", ["IndoorClimateObservations", "NoSuchObservations"], base)
    with pytest.raises(NotImplementedError):       # the reference cannot construct it either (observations.py:42)
        _check_supported("GreenhouseReward", ["IndoorClimateObservations", "StateObservations"], base)
    _check_supported("GreenhouseReward", OBSERVATION_MODULES, dict(base, delta_u_max=0.2))     # glgym_set_control_limits
    with pytest.raises(ValueError):
        _check_supported("GreenhouseReward", OBSERVATION_MODULES, dict(base, u_min=[0.5] * 6, u_max=[0.4] * 6))


@pytest.mark.gpu
def test_make_vec_env_from_reference_style_config(cfg_dir):
    """make_vec_env(env_id, env_base_params, env_specific_params, seed, n_envs, vec_norm_kwargs=...) end to end."""
    from gl_gym_amd.make_env import load_env_params, make_vec_env
    from gl_gym_amd.tomato_env import TomatoVecEnv
    base, spec = load_env_params("TomatoEnv", str(cfg_dir))
    from gl_gym_amd.vec_monitor import VecMonitorGPU
    mon_file = str(cfg_dir / "logs" / "train")
    env = make_vec_env("TomatoEnv", base, spec, seed=666, n_envs=256, monitor_filename=mon_file, dtype="float64")
    assert isinstance(env, VecMonitorGPU) and isinstance(env.venv, TomatoVecEnv)
    assert env.N == 48 and env.Np == 0 and env.num_envs == 256
    # what the experiment manager reads (RL/experiment_manager.py:43-45) and the reference's Box bounds
    mods = env.get_attr("observation_modules")[0]
    names = [n.replace("_", " ") for m in mods for n in m.obs_names]
    assert len(names) == env.obs_dim == 23 and names[:4] == ["co2 air", "temp air", "rh air", "pipe temp"]
    sp = env.observation_space
    assert np.all(sp.low[7:13] == 0) and np.all(sp.high[7:13] == 1) and sp.low[0] == np.float32(-1e-4) and sp.high[0] == 1e4
    obs = env.reset()
    days = np.array(env.get_attr("start_day"))
    assert set(np.unique(days)) == {0.0, 1.0} and 64 < (days == 0).sum() < 192          # both training days get drawn
    assert set(env.get_attr("growth_year")) == {2009}
    # an env that drew day 1 starts on the rows reset() would load for day 1
    b = int(np.nonzero(days == 1.0)[0][0])
    direct = TomatoVecEnv(4, weather=env.weather_data[env.start_rows[1]:], dt=900, season_length=0.5, pred_horizon=0.01,
                          dtype="float64", start_days=[1.0], auto_reset=False)
    np.testing.assert_array_equal(direct.reset()[0], obs[b])
    a = np.zeros((256, 6), np.float32)
    ret = np.zeros(256)
    for _ in range(49):
        obs, rew, done, infos = env.step(a)
        ret += rew
    assert done.all() and "terminal_observation" in infos[0]              # N + 1 = 49 steps, SB3 auto-reset
    # VecMonitor semantics: episode return = sum of raw rewards, length = N + 1; one CSV row per finished episode
    ep = infos[5]["episode"]
    assert ep["l"] == 49 and abs(ep["r"] - ret[5]) < 1e-5 and env.episode_count == 256
    assert float(env.episode_returns.abs().max()) == 0.0 and int(env.episode_lengths.max()) == 0
    obs, rew, done, infos = env.step(a)
    assert not done.any() and "episode" not in infos[5] and int(env.episode_lengths.min()) == 1
    # set_seed through env_method (experiments/evaluate_rl.py:113) re-keys the episode-start draws reproducibly
    env.env_method("set_seed", 777); env.reset(); d1 = np.array(env.get_attr("start_day"))
    env.env_method("set_seed", 777); env.reset(); d2 = np.array(env.get_attr("start_day"))
    env.env_method("set_seed", 778); env.reset(); d3 = np.array(env.get_attr("start_day"))
    assert np.array_equal(d1, d2) and not np.array_equal(d1, d3)
    env.close(); direct.close()
    rows = open(mon_file + ".monitor.csv").read().strip().split("\n")
    assert rows[0].startswith("#{") and rows[1] == "r,l,t" and len(rows) == 2 + 256
    # evaluation env with VecNormalize: statistics frozen, rewards raw (RL/utils.py:64-67)
    ev = make_vec_env("TomatoEnv", dict(base, training=False), spec, seed=1, n_envs=8, vec_norm_kwargs=dict(
        norm_obs=True, norm_reward=True, clip_obs=10.0, gamma=0.99), eval_env=True)
    assert ev.training is False and ev.norm_reward is False
    ev.reset()
    assert set(ev.get_attr("start_day")) == {1.0}                         # eval_options.eval_days
    for _ in range(49):
        obs, rew, done, infos = ev.step(np.zeros((8, 6), np.float32))
    assert done.all() and infos[0]["episode"]["l"] == 49                  # monitor entries survive the VecNormalize layer
    assert np.abs(infos[0]["terminal_observation"]).max() <= 10.0         # ... and terminal observations come normalised
    ev.close()
    # weather_on_device=True: the same table from the device pipeline (glgym_weather), never materialised on the host
    host_env = make_vec_env("TomatoEnv", base, spec, seed=666, n_envs=64, dtype="float64")
    dev_env = make_vec_env("TomatoEnv", base, spec, seed=666, n_envs=64, dtype="float64", weather_on_device=True)
    assert dev_env.weather_t.shape == host_env.weather_t.shape
    sc = host_env.weather_t.abs().amax(dim=0).clamp_min(1e-30)
    assert float(((dev_env.weather_t - host_env.weather_t).abs() / sc).max()) < 1e-12
    oh, od = host_env.reset(), dev_env.reset()
    np.testing.assert_allclose(od, oh, rtol=1e-6, atol=1e-6)
    for _ in range(3):
        rh_, rd_ = host_env.step(a[:64]), dev_env.step(a[:64])
        np.testing.assert_allclose(rd_[0], rh_[0], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(rd_[1], rh_[1], rtol=0, atol=1e-9)
    host_env.close(); dev_env.close()


def test_vecnormalize_reads_sb3_written_pickles_without_sb3(tmp_path):
    """The reference saves / loads SB3's own format (common/callbacks.py:292, experiment_manager.py:360,
    experiments/evaluate_rl.py:31): a pickled VecNormalize OBJECT whose state holds obs_rms / ret_rms
    (RunningMeanStd: mean, var, count), clip_obs, clip_reward, gamma, epsilon, norm_obs, norm_reward, training and the
    gymnasium spaces.  VecNormalizeGPU must read such a file on a box without stable_baselines3 / gymnasium (classes it
    cannot import are replaced by attribute bags), and its own files must round-trip."""
    import pickle
    import sys
    import types
    from gl_gym_amd.vec_normalize import VecNormalizeGPU
    # build an SB3-layout pickle with throw-away modules under SB3's real import paths
    names = ["stable_baselines3", "stable_baselines3.common", "stable_baselines3.common.vec_env",
             "stable_baselines3.common.vec_env.vec_normalize", "stable_baselines3.common.running_mean_std",
             "gymnasium", "gymnasium.spaces", "gymnasium.spaces.box"]
    mods = {n: types.ModuleType(n) for n in names}

    class RunningMeanStd:
        def __init__(self, shape=()):
            self.mean, self.var, self.count = np.zeros(shape), np.ones(shape), 1e-4
    RunningMeanStd.__module__ = "stable_baselines3.common.running_mean_std"
    RunningMeanStd.__qualname__ = "RunningMeanStd"

    class Box:
        def __init__(self):
            self.low, self.high, self.shape = np.zeros(3), np.ones(3), (3,)
    Box.__module__, Box.__qualname__ = "gymnasium.spaces.box", "Box"

    class VecNormalize:
        def __getstate__(self):            # SB3 drops venv / class_attributes / returns
            return {k: v for k, v in self.__dict__.items() if k not in ("venv", "returns")}
    VecNormalize.__module__ = "stable_baselines3.common.vec_env.vec_normalize"
    VecNormalize.__qualname__ = "VecNormalize"
    mods["stable_baselines3.common.running_mean_std"].RunningMeanStd = RunningMeanStd
    mods["stable_baselines3.common.vec_env.vec_normalize"].VecNormalize = VecNormalize
    mods["gymnasium.spaces.box"].Box = Box
    rng = np.random.default_rng(0)
    vn = VecNormalize()
    vn.obs_rms, vn.ret_rms = RunningMeanStd((263,)), RunningMeanStd(())
    vn.obs_rms.mean, vn.obs_rms.var, vn.obs_rms.count = rng.normal(size=263), rng.uniform(0.5, 2, 263), 12345.0
    vn.ret_rms.mean, vn.ret_rms.var, vn.ret_rms.count = 0.37, 2.5, 999.0
    vn.clip_obs, vn.clip_reward, vn.gamma, vn.epsilon = 10.0, 10.0, 0.9631, 1e-8
    vn.norm_obs, vn.norm_reward, vn.training = True, False, False
    vn.observation_space, vn.action_space, vn.venv, vn.returns = Box(), Box(), object(), np.zeros(8)
    saved = {n: sys.modules.get(n) for n in names}
    sys.modules.update(mods)
    try:
        path = tmp_path / "best_vecnormalize.pkl"
        with open(path, "wb") as f:
            pickle.dump(vn, f)
    finally:
        for n, m in saved.items():
            if m is None:
                sys.modules.pop(n, None)
            else:
                sys.modules[n] = m
    d = VecNormalizeGPU._read_stats(path)            # none of those modules is importable any more
    np.testing.assert_array_equal(d["obs_mean"], vn.obs_rms.mean)
    np.testing.assert_array_equal(d["obs_var"], vn.obs_rms.var)
    assert d["obs_count"] == 12345.0 and (d["ret_mean"], d["ret_var"], d["ret_count"]) == (0.37, 2.5, 999.0)
    assert d["gamma"] == 0.9631 and d["norm_reward"] is False and d["training"] is False and d["clip_obs"] == 10.0
    # the file VecNormalizeGPU.save writes: plain dict under the same attribute names
    own = tmp_path / "own.pkl"
    with open(own, "wb") as f:
        pickle.dump(dict(format="glgym-vecnormalize-2", obs_rms=dict(mean=vn.obs_rms.mean, var=vn.obs_rms.var, count=7.0),
                         ret_rms=dict(mean=0.1, var=0.2, count=3.0), clip_obs=5.0, clip_reward=10.0, gamma=0.99,
                         epsilon=1e-8, norm_obs=True, norm_reward=True, training=True), f)
    d2 = VecNormalizeGPU._read_stats(own)
    np.testing.assert_array_equal(d2["obs_var"], vn.obs_rms.var)
    assert d2["obs_count"] == 7.0 and d2["ret_var"] == 0.2 and d2["clip_obs"] == 5.0
    with open(tmp_path / "junk.pkl", "wb") as f:
        pickle.dump({"something": 1}, f)
    with pytest.raises(ValueError):
        VecNormalizeGPU._read_stats(tmp_path / "junk.pkl")
