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
    with pytest.raises(NotImplementedError):
        _check_supported("GreenhouseReward", OBSERVATION_MODULES[:-1], base)
    with pytest.raises(NotImplementedError):
        _check_supported("GreenhouseReward", OBSERVATION_MODULES, dict(base, delta_u_max=0.2))


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
