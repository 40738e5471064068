"""Config-driven construction with the reference's own signatures.

    load_env_params(env_id, path)                      <- gl_gym/common/utils.py:18-36
    make_vec_env(env_id, env_base_params, env_specific_params, seed, n_envs, monitor_filename, vec_norm_kwargs, eval_env)
                                                       <- gl_gym/RL/utils.py:44-69
    tomato_vec_env_from_config(n_envs, reward_function, observation_modules, constraints, eval_options,
                               reward_params, base_env_params, uncertainty_scale)
                                                       <- TomatoEnv.__init__ (gl_gym/environments/tomato_env.py:27-66) over
                                                          GreenLightEnv.__init__ (gl_gym/environments/base_env.py:40-88)

What changes versus the reference: every (growth_year, start_day) the env may draw at reset
(tomato_env.py:236-244: ``choice(train_years)``, ``choice(train_days)``; evaluation: ``eval_options``) is loaded ONCE
into one weather table resident in HBM, and an episode start becomes a row offset drawn on the device by
``glgym_reset``.  The reference re-reads and re-samples the CSV in every env at every reset.

The kernels implement the reference's GreenhouseReward, any list of its six live observation modules that starts with
IndoorClimateObservations (the reward reads obs[0:3]), and the yml's control limits (u_min, u_max, delta_u_max).  Anything
else (another reward class, other model sizes) is refused with NotImplementedError instead of being silently approximated.
"""
from __future__ import annotations

import os
from os.path import join
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .utils import load_weather_data

OBSERVATION_MODULES = ["IndoorClimateObservations", "BasicCropObservations", "ControlObservations", "WeatherObservations",
                       "TimeObservations", "WeatherForecastObservations"]          # TomatoEnv.yml:25-32, obs layout of glgym_obs


def load_env_params(env_id: str, path: str) -> Tuple[Dict, Dict]:
    """(env_base_params, env_specific_params) from ``<path>/<env_id>.yml`` -- same file format and return value as the
    reference's loader."""
    import yaml
    with open(join(path, env_id + ".yml"), "r") as f:
        params = yaml.load(f, Loader=yaml.FullLoader)
    env_specific_params = params[env_id] if env_id != "GreenLightEnv" else {}
    return params["GreenLightEnv"], env_specific_params


def season_table(weather_data_dir: str, location: str, data_source: str, years: Sequence[int], days: Sequence[float],
                 season_length: float, pred_horizon: float, dt: float, nd: int, pipeline=None):
    """One table holding the season window of every (year, start_day) pair, each loaded exactly as ``TomatoEnv.reset``
    does (tomato_env.py:249-259: ``load_weather_data(dir, location, source, year, start_day, season_length, Np + 1, dt, nd)``
    -- note the reference passes Np + 1 *steps* in the loader's *days* slot; kept, it only lengthens the window).
    Returns (table [rows, nd], start_rows, start_days, start_years).  With ``pipeline`` (a
    ``weather_device.WeatherPipeline``) the array work runs on the device and the table is a device tensor."""
    Np = int(pred_horizon * 86400 / dt)
    blocks, start_rows, start_days, start_years, row = [], [], [], [], 0
    for y in years:
        for d in days:
            load = load_weather_data if pipeline is None else pipeline.load_weather_data
            w = load(weather_data_dir, location, data_source, int(y), d, season_length, Np + 1, dt, nd)
            blocks.append(w)
            start_rows.append(row)
            start_days.append(float(d))
            start_years.append(int(y))
            row += len(w)
    table = np.concatenate(blocks, axis=0) if pipeline is None else pipeline.torch.cat(blocks, dim=0)
    return table, start_rows, start_days, start_years


def _check_supported(reward_function, observation_modules, base):
    if reward_function != "GreenhouseReward":
        raise NotImplementedError(f"reward_function {reward_function!r}: the step kernel implements GreenhouseReward "
                                  "(rewards.py:47-231) only")
    from .tomato_env import observation_modules as _modules
    _modules(0, observation_modules)      # any order / subset of the six live modules; raises for unknown names
    if (base.get("nx", 28), base.get("nu", 6), base.get("num_params", 208)) != (28, 6, 208) or not 10 <= base.get("nd", 10) <= 16:
        raise NotImplementedError("GreenLight sizes are nx=28, nu=6, nd=10..16, num_params=208")
    u_min, u_max = np.asarray(base.get("u_min", [0] * 6), float), np.asarray(base.get("u_max", [1] * 6), float)
    if u_min.shape != (6,) or u_max.shape != (6,) or np.any(u_min > u_max) or not float(base.get("delta_u_max", 0.1)) >= 0:
        raise ValueError("u_min / u_max need 6 entries with u_min <= u_max, delta_u_max >= 0")


def tomato_vec_env_from_config(n_envs: int, reward_function: str, observation_modules: List[str],
                               constraints: Dict[str, Any], eval_options: Dict[str, Any],
                               reward_params: Optional[Dict[str, Any]] = None,
                               base_env_params: Optional[Dict[str, Any]] = None, uncertainty_scale: float = 0.0,
                               seed: int = 0, **device_kw):
    """``TomatoEnv(**env_specific_params, base_env_params=env_base_params)`` for n_envs environments at once.
    device_kw: dtype, scheme, preset, n_sub, window, device, auto_reset, lazy_infos, model_variant (TomatoVecEnv keyword arguments);
    weather_on_device=True builds the weather table with the device pipeline (weather_device.py)."""
    from .tomato_env import TomatoVecEnv
    base = dict(base_env_params or {})
    _check_supported(reward_function, observation_modules, base)
    training = bool(base.get("training", True))
    if training:                                                         # base_env.py:81-82
        years = list(range(base.get("start_train_year", 2023), base.get("end_train_year", 2023) + 1))
        days = list(range(base.get("start_train_day", 265), base.get("end_train_day", 284) + 1))
        location, source = base["location"], base["data_source"]
    else:                                                                # tomato_env.py:240-244
        years, days = list(eval_options["eval_years"]), list(eval_options["eval_days"])
        location, source = eval_options["location"], eval_options["data_source"]
    dt, season, horizon = float(base.get("dt", 900)), base.get("season_length", 60), base.get("pred_horizon", 0.5)
    pipeline = None
    if device_kw.pop("weather_on_device", False):        # unit conversions, DLI / daylight flags, PCHIP in HIP kernels
        from .weather_device import WeatherPipeline
        pipeline = WeatherPipeline(device=device_kw.get("device", "cuda:0"), dtype="float64", nd=int(base.get("nd", 10)))
    table, rows, sdays, syears = season_table(base["weather_data_dir"], location, source, years, days, season, horizon,
                                              dt, int(base.get("nd", 10)), pipeline)
    if pipeline is not None:
        pipeline.close()
    env = TomatoVecEnv(n_envs, weather=table, dt=dt, season_length=season, pred_horizon=horizon, seed=seed,
                       start_rows=rows, start_days=sdays, reward_params=reward_params, constraints=constraints,
                       uncertainty_scale=uncertainty_scale, observation_modules=list(observation_modules),
                       u_min=base.get("u_min"), u_max=base.get("u_max"), delta_u_max=base.get("delta_u_max", 0.1), **device_kw)
    env.training, env.eval_options = training, eval_options
    env.location, env.data_source, env.weather_data_dir = location, source, base["weather_data_dir"]
    env.train_years, env.train_days = years, days
    env.start_years = np.asarray(syears)
    env.observation_module_names = list(observation_modules)
    return env


def make_vec_env(env_id: str, env_base_params: Dict[str, Any], env_specific_params: Dict[str, Any], seed: int,
                 n_envs: int, monitor_filename: Optional[str] = None, vec_norm_kwargs: Optional[Dict[str, Any]] = None,
                 eval_env: bool = False, **device_kw):
    """Same call as the reference's ``make_vec_env``; returns the B-env device environment instead of
    ``SubprocVecEnv([...] * n_envs)``, wrapped in the on-device VecNormalize when ``vec_norm_kwargs`` is given
    (evaluation envs: statistics frozen, rewards not normalised -- RL/utils.py:64-67).  The same three-layer stack as
    the reference: env -> VecMonitorGPU (episode return / length, ``infos[i]["episode"]``, optional monitor CSV) ->
    VecNormalizeGPU."""
    if env_id != "TomatoEnv":
        raise NotImplementedError(f"env_id {env_id!r}: only TomatoEnv exists (RL/utils.py:24)")
    env = tomato_vec_env_from_config(n_envs, base_env_params=env_base_params, seed=seed, **env_specific_params,
                                     **device_kw)
    from .vec_monitor import VecMonitorGPU
    if monitor_filename is not None and os.path.dirname(monitor_filename):
        os.makedirs(os.path.dirname(monitor_filename), exist_ok=True)
    env = VecMonitorGPU(env, filename=monitor_filename)
    if vec_norm_kwargs is not None:
        from .vec_normalize import VecNormalizeGPU
        env = VecNormalizeGPU(env, **vec_norm_kwargs)
        if eval_env:
            env.training = False
            env.norm_reward = False
    return env
