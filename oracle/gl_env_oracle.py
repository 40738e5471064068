"""CPU oracle for the environment semantics around the ODE step -- TEST INFRASTRUCTURE ONLY.

numpy restatement (single environment, fp64) of
  * TomatoEnv.reset / step / step_raw_control      gl_gym/environments/tomato_env.py:109-173,231-270
  * N, Np, bounds                                   gl_gym/environments/base_env.py:65-88
  * the six live observation modules                gl_gym/environments/observations.py:59-182
  * GreenhouseReward                                gl_gym/environments/rewards.py:96-124,156-231
  * parametric_crop_uncertainty                     gl_gym/environments/noise.py:3-23
  * init_state, co2dens2ppm, vaporPres2rh, satVp    gl_gym/environments/utils.py:13-46,281-309,364-379
Constants from gl_gym/configs/envs/TomatoEnv.yml.

Parity status: reward / noise / init_state / unit conversions are pinned by fixtures generated
from the reference's importable Python modules (tests/golden/{reward_kat,noise_draws,
params_default,weather_helpers}.npz); the sequencing itself is pinned only by the reference's
known-answer tests (tests/env_test.py:20-22,57,65,84-92), restated in tests/test_oracle_env.py.
"""
from __future__ import annotations

import numpy as np

from . import gl_oracle as O

INFO_KEYS = ["EPI", "revenue", "variable_costs", "fixed_costs", "co2_cost", "heat_cost", "elec_cost",
             "temp_violation", "co2_violation", "rh_violation", "lamp_violation"]

# configs/envs/TomatoEnv.yml:38-67
CONSTRAINTS_LOW = np.array([300.0, 15.0, 50.0])
CONSTRAINTS_HIGH = np.array([1600.0, 34.0, 85.0])
REWARD_DEFAULTS = dict(fixed_greenhouse_cost=15.0, fixed_co2_cost=0.015, fixed_lamp_cost=0.07,
                       fixed_screen_cost=2.0, elec_price=0.3, heating_price=0.09, co2_price=0.3,
                       fruit_price=1.6, dmfm=0.065, pen_weights=(4e-4, 5e-3, 7e-4), pen_lamp=0.1)


def sat_vp(t):                      # utils.py:281-291
    return 610.78 * np.exp(17.2694 * t / (t + 238.3))


def co2_dens_to_ppm(t, dens):       # utils.py:293-302 (same constants as aux_states.hpp:14-23)
    return 1e6 * 8.3144598 * (t + 273.15) * dens / (101325 * 44.01e-3)


def vapor_pres_to_rh(t, vp):        # utils.py:304-305
    return np.clip(100.0 * vp / sat_vp(t), 0.0, 100.0)


def init_state(d0, rh_max=90.0, time_in_days=0.0):   # utils.py:13-46
    x = np.zeros(28)
    t_air = 16.5
    x[0] = x[1] = d0[3]
    x[[2, 3, 5, 6, 7, 8, 9, 10, 17, 18, 19, 20]] = t_air
    x[4] = t_air + 4
    x[11] = 0.25 * (3.0 * t_air + d0[6])
    x[12] = 0.25 * (2.0 * t_air + 2 * d0[6])
    x[13] = 0.25 * (t_air + 3 * d0[6])
    x[14] = d0[6]
    x[15] = x[16] = rh_max / 100.0 * sat_vp(t_air)
    x[21] = x[4]
    x[22], x[23], x[24], x[25], x[26] = 0.0, 9.5283e4, 2.5107e5, 5.5338e4, 3.0978e3
    x[27] = time_in_days
    return x


def crop_noise(p32, scale, rng):    # noise.py:3-23 -- float32 arithmetic, 34 uniforms per call
    p = np.array(p32)
    idx = np.arange(128, 162)
    noise = rng.uniform(-scale / 2, scale / 2, size=idx.shape)
    p[idx] += noise * p[idx]
    p[144] = p[141] / p[142]
    return p


def gym_rng(seed):                  # gymnasium.utils.seeding.np_random
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))


class OracleReward:
    """rewards.py:47-231 restated (profit scaling + state-constraint violations)."""

    def __init__(self, env, constraints_low=None, constraints_high=None, **kw):
        k = dict(REWARD_DEFAULTS, **kw)
        self.env = env
        self.k = k
        self.low = CONSTRAINTS_LOW if constraints_low is None else np.asarray(constraints_low, dtype=np.float64)
        self.high = CONSTRAINTS_HIGH if constraints_high is None else np.asarray(constraints_high, dtype=np.float64)
        yearly = k["fixed_greenhouse_cost"] + k["fixed_co2_cost"] + k["fixed_lamp_cost"] * 116 + k["fixed_screen_cost"]
        self.fixed_costs = yearly / 365 / (86400 // env.dt)                       # :149-155
        p, dt = env.p, env.dt
        self.max_profit = p[154] * dt * 1e-6 / k["dmfm"] * k["fruit_price"]        # :96-106
        self.min_profit = -(p[108] / p[46] * dt / 3600 * 1e-3 * k["heating_price"]
                            + p[172] * dt / 3600 * 1e-3 * k["elec_price"]
                            + p[109] / p[46] * dt * 1e-6 * k["co2_price"])         # :108-124
        self.max_viol = np.array([2500.0, 15.0, 15.0])                            # :89-93
        self.profit = self.gains = self.variable_costs = 0.0
        self.heat_costs = self.co2_costs = self.elec_costs = 0.0
        self.temp_violation = self.co2_violation = self.rh_violation = self.lamp_violation = 0

    def compute_reward(self):
        e, k = self.env, self.k
        p, dt, u = e.p, e.dt, e.u
        self.heat_costs = u[0] * p[108] / p[46] * dt / 3600 * 1e-3 * k["heating_price"]   # :156-171
        self.elec_costs = u[4] * p[172] * dt / 3600 * 1e-3 * k["elec_price"]
        self.co2_costs = u[1] * p[109] / p[46] * dt * 1e-6 * k["co2_price"]
        self.variable_costs = self.heat_costs + self.co2_costs + self.elec_costs
        self.gains = (e.x[25] - e.x_prev[25]) * 1e-6 / k["dmfm"] * k["fruit_price"]          # :173-184
        self.profit = self.gains - self.variable_costs                                     # :221
        o = np.asarray(e.obs[:3], dtype=np.float64)
        viol = np.maximum(self.low - o, 0.0) + np.maximum(o - self.high, 0.0)                # :186-199
        self.co2_violation, self.temp_violation, self.rh_violation = viol
        self.lamp_violation = 0                                                            # :203-212 (always 0)
        scaled_profit = (self.profit - self.min_profit) / (self.max_profit - self.min_profit)
        return scaled_profit - np.sum(viol / self.max_viol) - self.lamp_violation * k["pen_lamp"]  # :228-231

    def info(self):
        return dict(EPI=self.profit, revenue=self.gains, variable_costs=self.variable_costs,
                    fixed_costs=self.fixed_costs, co2_cost=self.co2_costs, heat_cost=self.heat_costs,
                    elec_cost=self.elec_costs, temp_violation=self.temp_violation,
                    co2_violation=self.co2_violation, rh_violation=self.rh_violation,
                    lamp_violation=self.lamp_violation)


class OracleTomatoEnv:
    """Single-environment restatement of TomatoEnv (tomato_env.py) over the oracle step map."""

    def __init__(self, weather, p, season_length=60, start_day=59, growth_year=2010, dt=900.0,
                 pred_horizon=0.5, uncertainty_scale=0.0, integrator="rk4", n_sub=256, seed=None,
                 train_years=(2010,), train_days=(59,), reward_params=None, constraints=None,
                 observation_modules=None, u_min=None, u_max=None, delta_u_max=0.1):
        self.c = 86400
        self.nx, self.nu, self.nd, self.num_params = 28, 6, 10, 208
        self.dt = dt
        self.u_min = np.array([0.0] * 6 if u_min is None else u_min, dtype=np.float32)       # base_env.py:72-74
        self.u_max = np.array([1.0] * 6 if u_max is None else u_max, dtype=np.float32)
        self.delta_u_max = np.ones(6, dtype=np.float32) * delta_u_max
        self.Np = int(pred_horizon * self.c / dt)                  # base_env.py:80
        self.N = int(season_length * self.c / dt)                  # base_env.py:88
        self.weather_data = np.asarray(weather, dtype=np.float64)
        self.p = np.asarray(p, dtype=np.float32)
        self.uncertainty_scale = uncertainty_scale
        self.integrator, self.n_sub = integrator, n_sub
        self.constraints_low, self.constraints_high = CONSTRAINTS_LOW, CONSTRAINTS_HIGH
        self.train_years, self.train_days = list(train_years), list(train_days)
        self.start_day, self.growth_year = start_day, growth_year
        self.seed = seed
        # tomato_env.py:77-81: modules are concatenated in the order of the yml list
        self.observation_modules = list(observation_modules or ["IndoorClimateObservations", "BasicCropObservations",
                                                                "ControlObservations", "WeatherObservations",
                                                                "TimeObservations", "WeatherForecastObservations"])
        c = constraints or {}
        lo = [c.get("co2_min", 300.0), c.get("temp_min", 15.0), c.get("rh_min", 50.0)]
        hi = [c.get("co2_max", 1600.0), c.get("temp_max", 34.0), c.get("rh_max", 85.0)]
        self.constraints_low, self.constraints_high = np.array(lo), np.array(hi)
        self.reward = OracleReward(self, constraints_low=lo, constraints_high=hi, **(reward_params or {}))

    # greenlight_model.cpp:96-120 semantic
    def _evalF(self, x, u, d, p):
        p = np.asarray(p, dtype=np.float64)
        if self.integrator == "rk4":            # the kernels' scheme: stability-controlled RK4, Strang-split exact harvest
            # flow, tier 2b once per window of four sub-steps, guard retries
            return O.rk_sc_guarded(x, u, d, p, self.dt, self.n_sub, 4, getattr(self, "window", 4))[0]
        if self.integrator == "ls5":            # the kernels' five-stage fourth-order 2N scheme (gl_oracle.c ls5_substep), window 2
            return O.rk_sc_guarded(x, u, d, p, self.dt, self.n_sub, 5, getattr(self, "window", 2))[0]
        if self.integrator == "rk4_plain":      # classical RK4 of the complete RHS
            return O.rk4(x, u, d, p, self.dt, self.n_sub)
        if self.integrator == "stiff":
            return O.stiff(x, u, d, p, self.dt, 1e-10, 1e-10)[0]
        from scipy.integrate import solve_ivp
        tol = 1e-11 if self.integrator == "radau" else 1e-6
        s = solve_ivp(lambda t, y: O.rhs(y, u, d, p), (0.0, self.dt), x,
                      method="Radau" if self.integrator == "radau" else "BDF", rtol=tol, atol=tol)
        return s.y[:, -1]

    def reset(self, seed=None):                                     # tomato_env.py:231-270
        if seed is not None or not hasattr(self, "_np_random"):
            self._np_random = gym_rng(self.seed if seed is None else seed)
        self.growth_year = self._np_random.choice(self.train_years)
        self.start_day = self._np_random.choice(self.train_days)
        self.day_of_year = self.start_day
        self.hour_of_day = 0
        self.u = np.zeros(self.nu)
        self.x = init_state(self.weather_data[0])
        self.x_prev = np.copy(self.x)
        self.timestep = 0
        self.obs = self._get_obs()
        self.terminated = False
        return self.obs

    def action_to_control(self, action):                            # tomato_env.py:109-113
        return np.clip(self.u + action * self.delta_u_max, self.u_min, self.u_max)

    def _advance(self, params):
        self.x = self._evalF(self.x, self.u, self.weather_data[self.timestep], params)
        self.day_of_year += (self.dt / self.c) % 365                # tomato_env.py:126 (modulo on the increment)
        self.hour_of_day += self.dt / 3600
        self.hour_of_day = self.hour_of_day % 24
        self.obs = self._get_obs()
        if self.timestep >= self.N:                                 # tomato_env.py:68-75,131-132
            self.terminated = True

    def step(self, action, reward_hook=None):                       # tomato_env.py:115-146
        self.u = self.action_to_control(action)
        params = crop_noise(self.p, self.uncertainty_scale, self._np_random)
        self._advance(params)
        return self._finish(reward_hook)

    def step_raw_control(self, control, reward_hook=None):          # tomato_env.py:148-173
        self.u = np.asarray(control, dtype=np.float64)
        params = crop_noise(self.p, self.uncertainty_scale, self._np_random)
        self._advance(params)
        return self._finish(reward_hook)

    def _finish(self, reward_hook):
        r = self.reward.compute_reward()
        info = self.reward.info()
        if reward_hook is not None:       # fixture generation: the reference's GreenhouseReward bound to this env
            info["reward_ref"] = reward_hook.compute_reward()
            ref = dict(EPI=reward_hook.profit, revenue=reward_hook.gains, variable_costs=reward_hook.variable_costs,
                       fixed_costs=reward_hook.fixed_costs, co2_cost=reward_hook.co2_costs,
                       heat_cost=reward_hook.heat_costs, elec_cost=reward_hook.elec_costs,
                       temp_violation=reward_hook.temp_violation, co2_violation=reward_hook.co2_violation,
                       rh_violation=reward_hook.rh_violation, lamp_violation=reward_hook.lamp_violation)
            for k in INFO_KEYS:
                assert abs(ref[k] - info[k]) <= 1e-12 * max(1.0, abs(ref[k])), (k, ref[k], info[k])
        info["controls"] = self.u
        self.timestep += 1
        self.x_prev = np.copy(self.x)
        return self.obs, r, self.terminated, info

    def _get_obs(self):                                             # observations.py:70-182
        x, w, k = np.asarray(self.x), self.weather_data, self.timestep
        climate = np.array([co2_dens_to_ppm(x[2], x[0] * 1e-6), x[2], vapor_pres_to_rh(x[2], x[15]), x[9]])
        crop = x[[21, 25, 26]]
        row = w[k]
        weather = np.array([row[0], row[1], vapor_pres_to_rh(row[1], row[2]), co2_dens_to_ppm(row[1], row[3] * 1e-6),
                            row[4]])
        tm = np.array([k, np.sin(2 * np.pi * self.day_of_year / 365.0), np.cos(2 * np.pi * self.day_of_year / 365.0),
                       np.sin(2 * np.pi * self.hour_of_day / 24.0), np.cos(2 * np.pi * self.hour_of_day / 24.0)])
        forecast = w[k + 1:k + 1 + self.Np, 0:5].reshape(-1)      # raw rows, no unit conversion
        parts = {"IndoorClimateObservations": climate, "BasicCropObservations": crop,
                 "ControlObservations": np.asarray(self.u, dtype=np.float64), "WeatherObservations": weather,
                 "TimeObservations": tm, "WeatherForecastObservations": forecast}
        return np.concatenate([parts[m] for m in self.observation_modules])        # tomato_env.py:193-198
