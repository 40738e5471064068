"""Batched TomatoEnv on the MI355X.

``TomatoVecEnv`` keeps B independent greenhouse environments resident in HBM (state, controls, clocks,
weather window offsets) and advances all of them with ONE fused kernel launch per env-step
(libglgym.so: glgym_step), then assembles the row-major observation block (glgym_obs).  It exposes

  * the Stable-Baselines3 ``VecEnv`` calling convention that the reference's experiment manager consumes
    (gl_gym/RL/utils.py:44-69, gl_gym/common/evaluation.py:73-104): reset() / step(actions) -> (obs, rewards,
    dones, infos) with auto-reset and ``infos[i]["terminal_observation"]``, step_async/step_wait, num_envs,
    observation_space / action_space, get_attr / set_attr / env_method / env_is_wrapped / close;
  * a zero-copy tensor interface (``step_tensor``) for device-resident RL loops.

``TomatoEnv`` is the single-environment Gymnasium-style view (reset(seed) -> (obs, {}), step(a) ->
(obs, reward, terminated, False, info)) with the attributes the reference's tooling reads
(x, u, p, N, Np, timestep, weather_data, start_day, growth_year, ...; tomato_env.py:28-270).

Semantics reproduced from the reference (file:line in gl_gym/environments/):
  action -> control clip            tomato_env.py:109-113      episode = N+1 steps        tomato_env.py:131-137
  weather row = timestep (pre-++)   tomato_env.py:120          obs at step k uses row k   observations.py:133,161
  reward / info keys                rewards.py:218-231, tomato_env.py:208-222
  ODE failure -> terminated, x kept tomato_env.py:119-123
PyTorch is used for device memory, streams and RNG plumbing only; all arithmetic is in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from . import _lib as L
from .parameters import init_default_params
from .utils import synthetic_weather

try:  # spaces are optional plumbing: use gymnasium's when present
    from gymnasium import spaces as _spaces

    def _box(low, high, shape, dtype=np.float32):
        return _spaces.Box(low=low, high=high, shape=shape, dtype=dtype)
except Exception:  # pragma: no cover - gymnasium is absent in the build image
    class _Box:
        def __init__(self, low, high, shape, dtype):
            self.low = np.broadcast_to(np.asarray(low, dtype=dtype), shape).copy()
            self.high = np.broadcast_to(np.asarray(high, dtype=dtype), shape).copy()
            self.shape, self.dtype = tuple(shape), np.dtype(dtype)
            self._rng = np.random.default_rng()

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, v):
            v = np.asarray(v)
            return v.shape == self.shape and bool(np.all(v >= self.low) and np.all(v <= self.high))

    def _box(low, high, shape, dtype=np.float32):
        return _Box(low, high, shape, dtype)

DEFAULT_REWARD = dict(fixed_greenhouse_cost=15., fixed_co2_cost=0.015, fixed_lamp_cost=0.07, fixed_screen_cost=2.,
                      elec_price=0.3, heating_price=0.09, co2_price=0.3, fruit_price=1.6, dmfm=0.065,
                      pen_weights=[4.e-4, 5.e-3, 7.e-4], pen_lamp=0.1)               # TomatoEnv.yml:56-67
DEFAULT_CONSTRAINTS = dict(co2_min=300., co2_max=1600., temp_min=15., temp_max=34., rh_min=50., rh_max=85.)


def _torch():
    import torch
    return torch


def _observation_modules(Np, names):       # TomatoVecEnv.__init__ has a keyword argument of the same name as the function
    return observation_modules(Np, names)


class ObservationModule:
    """Descriptor of one of the reference's observation modules (observations.py:59-182): what the experiment manager
    reads through ``env.get_attr("observation_modules")`` (RL/experiment_manager.py:43-45).  The values themselves are
    produced by glgym_obs; ``low`` / ``high`` are the module's Box bounds as the reference declares them."""

    def __init__(self, name, obs_names, low=-1e-4, high=1e4):
        self.name, self.obs_names, self.n_obs, self.low, self.high = name, list(obs_names), len(obs_names), low, high

    def __repr__(self):
        return f"{self.name}({self.n_obs})"


OBSERVATION_MODULE_IDS = {"IndoorClimateObservations": 0, "BasicCropObservations": 1, "ControlObservations": 2,
                          "WeatherObservations": 3, "TimeObservations": 4, "WeatherForecastObservations": 5}   # GLGYM_OBS_*
DEFAULT_OBSERVATION_MODULES = ["IndoorClimateObservations", "BasicCropObservations", "ControlObservations",
                               "WeatherObservations", "TimeObservations", "WeatherForecastObservations"]  # TomatoEnv.yml:25-32


def observation_modules(Np: int, names: Optional[Sequence[str]] = None):
    """Descriptors of the configured modules, in the order of the list = the column layout of glgym_obs
    (tomato_env.py:77-81, 193-198).  Default: the six of configs/envs/TomatoEnv.yml."""
    wx = ["glob_rad", "temp_out", "rh_out", "co2_out", "wind_speed"]
    table = {"IndoorClimateObservations": (["co2_air", "temp_air", "rh_air", "pipe_temp"],),
             "BasicCropObservations": (["24CanTemp", "cFruit", "tSum"],),
             "ControlObservations": (["uBoil", "uCo2", "uThScr", "uVent", "uLamp", "uBlScr"], 0.0, 1.0),
             "WeatherObservations": (wx,),
             "TimeObservations": (["timestep", "day of year sin", "day of year cos", "hour of day sin",
                                   "hour of day cos"],),
             "WeatherForecastObservations": (wx * Np,)}
    names = DEFAULT_OBSERVATION_MODULES if names is None else list(names)
    for n in names:
        if n == "StateObservations":
            raise NotImplementedError("StateObservations cannot be built by the reference's TomatoEnv either "
                                      "(observations.py:42: __init__ takes no env) and returns random numbers (:57)")
        if n not in table:
            raise KeyError(f"unknown observation module {n!r}; known: {sorted(table)}")
    if len(set(names)) != len(names) or not names:
        raise ValueError("observation_modules must be a non-empty list without repeats")
    return [ObservationModule(n, *table[n]) for n in names]


class LazyInfos:
    """List-like view of the per-env info dicts (SB3: ``infos[i]``).  Dicts are built on first access and cached, so
    wrappers that only touch finished envs (VecMonitor, VecNormalize) cost O(#done), not O(B), per step."""

    def __init__(self, keys, info_rows, controls, dones, term_obs, step_flags=None):
        self._keys, self._rows, self._ctrl, self._dones, self._term = keys, info_rows, controls, dones, term_obs
        self._flags = step_flags
        self._cache = {}

    def __len__(self):
        return len(self._rows)

    def _make(self, b):
        d = dict(zip(self._keys, self._rows[b].tolist()))
        d["controls"] = self._ctrl[b]
        d["TimeLimit.truncated"] = False
        if self._flags is not None:
            d["integration"] = int(self._flags[b])        # GLGYM_SF_* word: 0 = first attempt accepted as it stood
        if self._term is not None and self._dones[b]:
            d["terminal_observation"] = self._term[b]
        return d

    def __getitem__(self, b):
        if isinstance(b, slice):
            return [self[i] for i in range(*b.indices(len(self)))]
        b = int(b)
        if b < 0:
            b += len(self)
        if b not in self._cache:
            self._cache[b] = self._make(b)
        return self._cache[b]

    def __setitem__(self, b, value):
        self._cache[int(b)] = value

    def __iter__(self):
        return (self[i] for i in range(len(self)))


class TomatoVecEnv:
    def __init__(self, num_envs: int, weather: Optional[np.ndarray] = None, params: Optional[np.ndarray] = None,
                 dt: float = 900.0, season_length: float = 60, pred_horizon: float = 0.5, dtype: str = "float32",
                 n_sub: Optional[int] = None, device: str = "cuda:0", seed: int = 0, uncertainty_scale: float = 0.0,
                 start_rows: Optional[Sequence[int]] = None, start_days: Optional[Sequence[float]] = None,
                 reward_params: Optional[Dict[str, Any]] = None, constraints: Optional[Dict[str, float]] = None,
                 auto_reset: bool = True, collect_metrics: bool = True, lazy_infos: Optional[bool] = None,
                 model_variant: str = "ode", scheme: Optional[str] = None, window: Optional[int] = None,
                 preset: Optional[str] = None,
                 observation_modules: Optional[Sequence[str]] = None, u_min: Optional[Sequence[float]] = None,
                 u_max: Optional[Sequence[float]] = None, delta_u_max: float = 0.1, integration_info: bool = True):
        """u_min / u_max / delta_u_max: action_to_control's bounds (base_env.py:72-74; default [0, 1] and 0.1).
        observation_modules: names of the reference's modules in output order (default: the six of TomatoEnv.yml).
        scheme / n_sub / window: "ls5" (default; model_variant "ode_pipe" defaults to "rk4" and accepts no other: five-stage fourth-order 2N-storage scheme, n_sub 128, two sub-steps per tier-2b window), "rk4" (classical RK4, 240),
        "rk3" (three-stage third-order scheme, 270) or "rk2" (midpoint rule, 336), all with the cover conduction integrated exactly (include/glgym.h).
        preset: "throughput" (the counts above; default for float32) or "parity" (inside the band of the reference solver's tolerances:
        ls5 n_sub 192 with one sub-step per window; default for float64) -- used for whatever of n_sub / window is not given (_lib.PRESETS);
        integration_info: put the per-env GLGYM_SF_* word of each step into infos[i]["integration"] (one more B x 4-byte D2H copy
        per host-side step; False: not copied, key absent -- the device tensor `step_flags_t` is always written);
        weather: [rows, nd] with nd = 10, or 14 when the rows carry the measured pipe columns
        (experiments/gl_predefined_controls.py:95, 107).  model_variant = "ode" | "ode_pipe" (ode.hpp:126-263, nd >= 14)."""
        torch = _torch()
        if not torch.cuda.is_available():
            raise L.GlgymError("TomatoVecEnv needs a HIP device (torch.cuda.is_available() is False); "
                               "there is no CPU fallback")
        self._lib = L.load()
        self.torch = torch
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.num_envs = self.B = int(num_envs)
        self.ld = (self.B + 63) // 64 * 64
        self.dt = float(dt)
        self.c = 86400
        self.nx, self.nu, self.num_params = L.NX, L.NU, L.NP
        weather_is_tensor = weather is not None and hasattr(weather, "data_ptr")          # a torch tensor (device table)
        self.nd = L.ND if weather is None else int(np.shape(weather)[1])        # arrays and nested lists alike
        if model_variant not in ("ode", "ode_pipe"):
            raise ValueError("model_variant must be 'ode' or 'ode_pipe'")
        self.model_variant = model_variant
        self.season_length = season_length
        self.N = int(season_length * self.c / self.dt)                     # base_env.py:88
        self.Np = int(pred_horizon * self.c / self.dt)                     # base_env.py:80
        self.observation_modules = _observation_modules(self.Np, observation_modules)
        if self.observation_modules[0].name != "IndoorClimateObservations":
            # GreenhouseReward reads its constraint inputs positionally, obs[[0, 1, 2]] = co2_air, temp_air, rh_air
            # (rewards.py:192-194): with another module first the reference penalises whatever lands in those slots
            raise NotImplementedError("observation_modules must start with IndoorClimateObservations: the reference's "
                                      "GreenhouseReward takes co2 / temperature / humidity from obs[0:3]")
        self.obs_dim = sum(m.n_obs for m in self.observation_modules)
        self.f64 = str(dtype) in ("float64", "f64", "double")
        self.tdtype = torch.float64 if self.f64 else torch.float32
        scheme = L.resolve_scheme(scheme, model_variant)     # None: "ls5"; "rk4" for ode_pipe (the only scheme its kernels are built for)
        self.preset = ("parity" if self.f64 else "throughput") if preset is None else preset
        if self.preset not in L.PRESETS:
            raise ValueError("preset must be 'throughput' or 'parity'")
        self.scheme = scheme
        n_def, w_def = L.preset_n_sub(scheme, self.dt, self.preset)
        self.n_sub = int(n_def if n_sub is None else n_sub)
        self.window = int((w_def if n_sub is None else 0) if window is None else window)      # 0 = the scheme's own
        self.uncertainty_scale = float(uncertainty_scale)
        self.auto_reset = auto_reset
        self.lazy_infos = (int(num_envs) > 4096) if lazy_infos is None else bool(lazy_infos)
        self.integration_info = bool(integration_info)
        self.seed_value = int(seed)

        self._p = np.asarray(init_default_params(L.NP) if params is None else params, dtype=np.float32)
        self._h = C.c_void_p()
        p64 = np.ascontiguousarray(self.p, dtype=np.float64)
        L.check(self._lib.glgym_create(L.NX, L.NU, self.nd, L.NP, self.dt, p64.ctypes.data_as(L._DP),
                                       L.F64 if self.f64 else L.F32, self.n_sub, self.device.index or 0,
                                       C.byref(self._h)), "glgym_create")
        # GLGYM_LAYOUT / GLGYM_OCC / GLGYM_VERIFY are read ONCE, here in glgym_create (handle state since round 5): changing os.environ
        # afterwards has no effect on this handle -- _launch_step warns once if that is attempted (use set_layout / set_occupancy / set_verify)
        self._env_at_create = tuple(os.environ.get(k) for k in ("GLGYM_LAYOUT", "GLGYM_OCC", "GLGYM_VERIFY"))
        if model_variant == "ode_pipe":
            L.check(self._lib.glgym_set_model_variant(self._h, L.ODE_PIPE), "glgym_set_model_variant")
        L.check(self._lib.glgym_set_scheme(self._h, L.SCHEMES[scheme]), "glgym_set_scheme")
        L.check(self._lib.glgym_set_window(self._h, self.window), "glgym_set_window")
        self.u_min = np.asarray([0.0] * L.NU if u_min is None else u_min, dtype=np.float32)            # base_env.py:72-74
        self.u_max = np.asarray([1.0] * L.NU if u_max is None else u_max, dtype=np.float32)
        self.delta_u_max = np.ones(L.NU, dtype=np.float32) * np.float32(delta_u_max)
        if self.u_min.shape != (L.NU,) or self.u_max.shape != (L.NU,):
            raise ValueError("u_min / u_max must have 6 entries")
        lo64, hi64 = self.u_min.astype(np.float64), self.u_max.astype(np.float64)
        L.check(self._lib.glgym_set_control_limits(self._h, lo64.ctypes.data_as(L._DP), hi64.ctypes.data_as(L._DP),
                                                   float(delta_u_max)), "glgym_set_control_limits")
        ids = (C.c_int32 * len(self.observation_modules))(*[OBSERVATION_MODULE_IDS[m.name] for m in self.observation_modules])
        L.check(self._lib.glgym_set_obs_modules(self._h, ids, len(ids)), "glgym_set_obs_modules")
        assert self._lib.glgym_obs_dim(self._h, self.Np) == self.obs_dim
        rp = dict(DEFAULT_REWARD, **(reward_params or {}))
        cs = dict(DEFAULT_CONSTRAINTS, **(constraints or {}))
        self.reward_params, self.constraints = rp, cs
        cfg = L.RewardCfg(rp["elec_price"], rp["heating_price"], rp["co2_price"], rp["fruit_price"], rp["dmfm"],
                          rp["fixed_greenhouse_cost"], rp["fixed_co2_cost"], rp["fixed_lamp_cost"],
                          rp["fixed_screen_cost"], rp["pen_lamp"], cs["co2_min"], cs["co2_max"], cs["temp_min"],
                          cs["temp_max"], cs["rh_min"], cs["rh_max"])
        L.check(self._lib.glgym_set_reward(self._h, C.byref(cfg)), "glgym_set_reward")
        mx, mn, fx = C.c_double(), C.c_double(), C.c_double()
        L.check(self._lib.glgym_get_reward_scale(self._h, C.byref(mx), C.byref(mn), C.byref(fx)))
        self.max_profit, self.min_profit, self.fixed_costs = mx.value, mn.value, fx.value

        # ---- weather tensor (shared by all envs) and the admissible episode start rows
        if weather is None:
            weather = synthetic_weather(dt=self.dt)
        self._weather_data = (weather.detach().double().cpu().numpy() if weather_is_tensor
                              else np.ascontiguousarray(weather, dtype=np.float64))
        self.weather_rows = int(self._weather_data.shape[0])
        need = self.N + 1 + self.Np + 1
        if self.weather_rows < need:
            raise ValueError(f"weather tensor has {self.weather_rows} rows, an episode needs {need}")
        if start_rows is None:
            start_rows = [0]
        self.start_rows = np.asarray(start_rows, dtype=np.int64)
        if np.any(self.start_rows + need > self.weather_rows) or np.any(self.start_rows < 0):
            raise ValueError("start_rows + episode length exceeds the weather tensor")
        self.start_days = np.asarray(start_days if start_days is not None
                                     else self.start_rows * self.dt / self.c, dtype=np.float32)

        dev, T = self.device, self.tdtype
        z = lambda *s, dtype=T: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self.weather_t = (weather.to(device=dev, dtype=T).contiguous() if weather_is_tensor
                          else torch.as_tensor(self.weather_data, dtype=T, device=dev).contiguous())
        self.x_T, self.u_T = z(L.NX, self.ld), z(L.NU, self.ld)
        self.ctrl_T = z(L.NU, self.ld)
        self.info_T = z(L.NINFO, self.ld)
        self.reward_t = z(self.ld)
        self.done_t = z(self.B, dtype=torch.uint8)
        self.timestep_t = z(self.B, dtype=torch.int32)
        self.w_off_t = z(self.B, dtype=torch.int32)
        self.start_day_t = z(self.B, dtype=torch.float32)
        self.action_t = z(self.B, L.NU, dtype=torch.float32)
        self.obs_t = z(self.B, self.obs_dim, dtype=torch.float32)
        self.term_obs_t = z(self.B, self.obs_dim, dtype=torch.float32)
        self.metrics_t = z(L.METRIC_REPLICAS, L.METRIC_STRIDE, dtype=torch.float32) if collect_metrics else None
        # how each env-step's integration went (include/glgym.h GLGYM_SF_*: first-attempt flags, extra attempts, and whether the
        # result was accepted by agreement on a flagged attempt / as the finest attempt alone / not at all); infos["integration"]
        self.step_flags_t = z(self.B, dtype=torch.int32)
        self.crop_T = z(L.NCROP, self.ld) if self.uncertainty_scale > 0 else None
        self._start_rows_t = torch.as_tensor(self.start_rows, dtype=torch.int32, device=dev)
        self._start_days_t = torch.as_tensor(self.start_days, dtype=torch.float32, device=dev)
        self.episode_t = z(self.B, dtype=torch.int32)           # episodes started per env (keys the start draw)
        self._draw = 0
        self.x, self.u = self.x_T[:, :self.B].t(), self.u_T[:, :self.B].t()      # [B,28] / [B,6] views

        # observation space = concatenation of the modules' bounds (tomato_env.py:83-95; the reference's lower bound
        # really is -1e-4)
        lo = np.concatenate([np.full(m.n_obs, m.low, np.float32) for m in self.observation_modules])
        hi = np.concatenate([np.full(m.n_obs, m.high, np.float32) for m in self.observation_modules])
        self.observation_space = _box(lo, hi, (self.obs_dim,), np.float32)
        self.action_space = _box(-1.0, 1.0, (L.NU,), np.float32)
        self._actions = None
        self._keep_applied_u, self._u_applied_T = False, None
        self.reset_infos: List[dict] = [{} for _ in range(self.B)]
        # pinned host staging for the numpy (SB3) path: the 1 KB/env observation block dominates the D2H traffic
        self._obs_host = [torch.empty(self.B, self.obs_dim, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._obs_flip = 0

    def _obs_to_host(self, obs_t):
        """D2H into one of two pinned buffers (alternating) and return a numpy VIEW of it: the array stays valid
        until the call after next.  (A second host-side copy of the 1 KB/env block would cost more than the PCIe
        transfer itself; consumers such as SB3's rollout buffer copy what they keep.)"""
        buf = self._obs_host[self._obs_flip]
        self._obs_flip ^= 1
        buf.copy_(obs_t, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return buf.numpy()

    # ------------------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _launch_reset(self, mask_t):
        """Masked reset; the kernel draws each new episode's start from the start table (Philox on (seed, env, episode))."""
        a = L.ResetArgs(self.B, self.ld, mask_t.data_ptr() if mask_t is not None else None, self.x_T.data_ptr(),
                        self.u_T.data_ptr(), self.timestep_t.data_ptr(), self.weather_t.data_ptr(), self.weather_rows,
                        self.w_off_t.data_ptr(), self._start_rows_t.data_ptr(), self._start_days_t.data_ptr(),
                        len(self.start_rows), self.start_day_t.data_ptr(), self.episode_t.data_ptr(), self.seed_value)
        L.check(self._lib.glgym_reset(self._h, C.byref(a), self._stream()), "glgym_reset")

    def _launch_obs(self, out_t, mask_t=None, term_t=None):
        a = L.ObsArgs(self.B, self.ld, self.x_T.data_ptr(), self.u_T.data_ptr(), self.weather_t.data_ptr(),
                      self.weather_rows, self.w_off_t.data_ptr(), self.timestep_t.data_ptr(),
                      self.start_day_t.data_ptr(), self.Np, out_t.data_ptr(),
                      mask_t.data_ptr() if mask_t is not None else None,
                      term_t.data_ptr() if term_t is not None else None)
        L.check(self._lib.glgym_obs(self._h, C.byref(a), self._stream()), "glgym_obs")

    def _launch_step(self, raw_control: bool):
        if self._env_at_create is not None and tuple(os.environ.get(k) for k in ("GLGYM_LAYOUT", "GLGYM_OCC", "GLGYM_VERIFY")) != self._env_at_create:
            import warnings
            warnings.warn("GLGYM_LAYOUT / GLGYM_OCC / GLGYM_VERIFY changed after this TomatoVecEnv was created: they are read once, at glgym_create, "
                          "and have no effect on an existing handle -- use set_layout() / set_occupancy() / set_verify()", RuntimeWarning, stacklevel=3)
            self._env_at_create = None                     # once
        if self.crop_T is not None and not getattr(self, "freeze_crop_noise", False):       # noise.py: a fresh draw every step
            L.check(self._lib.glgym_crop_noise(self._h, self.crop_T.data_ptr(), self.B, self.ld,
                                               self.uncertainty_scale, self.seed_value, self._draw, self._stream()),
                    "glgym_crop_noise")
            self._draw += 1
        a = L.make_step_args(self.B, self.ld, self.x_T.data_ptr(), self.u_T.data_ptr(),
                       None if raw_control else getattr(self, "_action_src", self.action_t).data_ptr(),
                       self.ctrl_T.data_ptr() if raw_control else None, self.weather_t.data_ptr(), self.weather_rows,
                       self.w_off_t.data_ptr(), self.timestep_t.data_ptr(),
                       self.crop_T.data_ptr() if self.crop_T is not None else None, self.N,
                       self.reward_t.data_ptr(), self.info_T.data_ptr(), self.done_t.data_ptr(),
                       self.metrics_t.data_ptr() if self.metrics_t is not None else None, self.step_flags_t.data_ptr())
        L.check(self._lib.glgym_step(self._h, C.byref(a), self._stream()), "glgym_step")

    # ---- tensor interface (no host synchronisation) ---------------------------------------------
    def reset_tensor(self, seed: Optional[int] = None):
        if seed is not None:
            self.seed_value = int(seed)
            self.episode_t.zero_()
        self._launch_reset(None)
        self._launch_obs(self.obs_t)
        return self.obs_t

    def _launch_rule_based(self, controller, hour_t=None, doy_t=None):
        """glgym_rule_based: controller.predict for every env, written to the SoA control buffer of step_raw_control."""
        cfg = L.RuleCfg(*[float(getattr(controller, n)) for n in L.RULE_FIELDS])
        if hour_t is None and (self.dt / 3600) * 8 != round((self.dt / 3600) * 8):    # increments other than multiples of 1/8 h do not sum exactly
            hour_t = self.hour_of_day()
        a = L.RuleArgs(self.B, self.ld, self.x_T.data_ptr(), self.weather_t.data_ptr(), self.weather_rows,
                       self.w_off_t.data_ptr(), self.timestep_t.data_ptr(), self.start_day_t.data_ptr(),
                       hour_t.data_ptr() if hour_t is not None else None,
                       doy_t.data_ptr() if doy_t is not None else None, self.ctrl_T.data_ptr())
        L.check(self._lib.glgym_rule_based(self._h, C.byref(cfg), C.byref(a), self._stream()), "glgym_rule_based")

    def step_tensor(self, actions_t=None, controls_t=None, want_obs: bool = True, controller=None):
        """actions_t [B,6] f32 in [-1,1] (step), controls_t [B,6] (step_raw_control) or controller (a
        RuleBasedController evaluated on the device, then step_raw_control).  Returns
        (obs [B,dim] f32, reward [B], done [B] uint8, info [11,B]) as device tensors; with auto_reset the
        finished envs are re-initialised and ``term_obs_t`` keeps their last observation."""
        if (actions_t is not None) + (controls_t is not None) + (controller is not None) != 1:
            raise ValueError("give exactly one of actions_t / controls_t / controller")
        self._action_src = self.action_t
        if actions_t is not None:
            if (actions_t.dtype == self.torch.float32 and actions_t.is_cuda and actions_t.device == self.device and
                    actions_t.is_contiguous() and actions_t.numel() == self.B * L.NU):
                self._action_src = actions_t          # read in place by the kernel (no 1.5 MB copy, no extra launch)
            else:
                self.action_t.copy_(actions_t.reshape(self.B, L.NU))
        elif controls_t is not None:
            self.ctrl_T[:, :self.B].copy_(controls_t.reshape(self.B, L.NU).t())
        else:
            self._launch_rule_based(controller)
        self._launch_step(raw_control=actions_t is None)
        if want_obs:
            self._launch_obs(self.obs_t)
        if self.auto_reset:      # SB3 semantics: finished envs restart; their last obs goes to term_obs_t
            if self._keep_applied_u:     # host infos report the control applied in THIS step (tomato_env.py:221), which the
                if self._u_applied_T is None:                        # reset below zeroes for the finished envs
                    self._u_applied_T = self.torch.empty_like(self.u_T)
                self._u_applied_T.copy_(self.u_T)
                self._u_applied_valid = True
            else:
                self._u_applied_valid = False        # a later host_infos() must not report an EARLIER step's controls
            self._launch_reset(self.done_t)
            if want_obs:
                self._launch_obs(self.obs_t, self.done_t, self.term_obs_t)
        return self.obs_t, self.reward_t[:self.B], self.done_t, self.info_T[:, :self.B]

    # ---- SB3 VecEnv calling convention ------------------------------------------------------------
    def reset(self):
        return self._obs_to_host(self.reset_tensor())

    def seed(self, seed: Optional[int] = None):
        if seed is not None:
            self.seed_value = int(seed)
        return [seed] * self.B

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        self._keep_applied_u = True
        try:
            return self._host_result(self.step_tensor(self.torch.as_tensor(self._actions, device=self.device)))
        finally:
            self._keep_applied_u = False

    def _host_result(self, out):
        """(obs, rewards, dones, infos) as SB3 consumes them, from step_tensor's device tensors."""
        obs_t, r_t, d_t, info_T = out
        obs, rew = self._obs_to_host(obs_t), r_t.float().cpu().numpy()
        dones, infos = self.host_infos(d_t, info_T)
        return obs, rew, dones, infos

    def host_infos(self, d_t, info_T, term_obs=None):
        """(dones [B] bool, infos) on the host.  term_obs: replacement for the raw terminal observations (a wrapper
        that rescales observations passes its own)."""
        dones = d_t.cpu().numpy().astype(bool)
        # infos: SB3 wants a list of per-env dicts.  Built from two bulk D2H copies (info block, controls) with
        # zip over Python lists -- the cheapest pure-Python construction (about 1 us per env per key).
        rows = info_T.double().t().cpu().numpy()
        applied = self._u_applied_T if (self.auto_reset and getattr(self, "_u_applied_valid", False)) else self.u_T
        ctrl = applied[:, :self.B].t().double().cpu().numpy()
        term = None
        if self.auto_reset and dones.any():
            term = self.term_obs_t.cpu().numpy() if term_obs is None else term_obs
        flags = self.step_flags_t.cpu().numpy() if self.integration_info else None
        if self.lazy_infos:
            infos = LazyInfos(L.INFO_KEYS, rows, ctrl, dones, term, flags)
        else:
            infos = [dict(zip(L.INFO_KEYS, row), controls=c) for row, c in zip(rows.tolist(), ctrl)]
            for d in infos:
                d["TimeLimit.truncated"] = False
            if flags is not None:
                for d, f in zip(infos, flags.tolist()):
                    d["integration"] = f      # GLGYM_SF_* word: 0 = first attempt accepted as it stood
            if term is not None:
                for b in np.nonzero(dones)[0]:
                    infos[b]["terminal_observation"] = term[b]
        return dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def step_raw_control(self, controls):
        torch = self.torch
        obs_t, r_t, d_t, info_T = self.step_tensor(controls_t=torch.as_tensor(np.asarray(controls), dtype=self.tdtype,
                                                                              device=self.device))
        return self._obs_to_host(obs_t), r_t.float().cpu().numpy(), d_t.cpu().numpy().astype(bool), info_T.cpu().numpy()

    def step_raw_control_pipeinput(self, controls):
        """tomato_env.py:175-191: controls applied directly, no noise / observation / info; returns (x [B,28], terminated [B])."""
        torch = self.torch
        _, _, d_t, _ = self.step_tensor(controls_t=torch.as_tensor(np.asarray(controls), dtype=self.tdtype,
                                                                   device=self.device), want_obs=False)
        return self.x.double().cpu().numpy(), d_t.cpu().numpy().astype(bool)

    def _indices(self, indices):
        if indices is None:
            return list(range(self.B))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name: str, indices=None):
        idx = self._indices(indices)
        per_env = {"x": lambda b: self.x[b].double().cpu().numpy(), "u": lambda b: self.u[b].double().cpu().numpy(),
                   "timestep": lambda b: int(self.timestep_t[b]), "start_day": lambda b: float(self.start_day_t[b]),
                   "w_off": lambda b: int(self.w_off_t[b])}
        if attr_name == "growth_year" and getattr(self, "start_years", None) is not None:
            # the (year, day) pair an env drew at its last reset = the entry of the start table its w_off points at
            w_off = self.w_off_t.cpu().numpy()
            return [int(self.start_years[int(np.searchsorted(self.start_rows, w_off[b], side="right")) - 1]) for b in idx]
        if attr_name in per_env:
            return [per_env[attr_name](b) for b in idx]
        return [getattr(self, attr_name) for _ in idx]

    def set_attr(self, attr_name: str, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name: str, *args, indices=None, **kwargs):
        return [getattr(self, method_name)(*args, **kwargs) for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_obs_names(self):
        return [n for m in self.observation_modules for n in m.obs_names]

    def set_seed(self, seed: int):
        """base_env.py:166-170 (called through env_method in experiments/evaluate_rl.py:113): re-seed the generator
        that draws the episode starts."""
        self.seed_value = int(seed)
        self.episode_t.zero_()

    # ---- clocks and the weather row of the coming step, as device tensors (for controllers) ---------
    def current_weather(self):
        """[B,10] rows weather[w_off + timestep] -- what the reference passes to RuleBasedController.predict."""
        idx = (self.w_off_t + self.timestep_t).long().clamp_(0, self.weather_rows - 1)
        return self.weather_t[idx]

    @property
    def weather_data(self):
        """The resident weather table [rows, nd] (float64 copy on the host).  ASSIGNING to it is the reference's `env.weather_data = table` on a
        constructed env (experiments/run_time.py:39, gl_predefined_controls.py:114): the reference's step reads `weather_data[timestep]`, so the new
        table is indexed from its row 0 by every environment, running episodes included (their `timestep` is kept), and later resets start at row
        0 / the first start day.  The column count is fixed at construction (`nd`: 10, or 14 for the measured-pipe variant); a step graph captured before
        the assignment holds the old table's address and has to be captured again."""
        return self._weather_data

    @weather_data.setter
    def weather_data(self, table):
        torch = self.torch
        table = np.ascontiguousarray(table.detach().double().cpu().numpy() if hasattr(table, "detach") else table, dtype=np.float64)
        if table.ndim != 2 or table.shape[1] != self._weather_data.shape[1]:
            raise ValueError(f"weather_data: expected [rows, {self._weather_data.shape[1]}] like the table this env was built with, got {table.shape}")
        need = self.N + 1 + self.Np + 1
        if table.shape[0] < need:
            raise ValueError(f"weather_data has {table.shape[0]} rows, an episode needs {need}")
        self._weather_data = table
        self.weather_rows = int(table.shape[0])
        self.weather_t = torch.as_tensor(table, dtype=self.tdtype, device=self.device).contiguous()
        self.start_rows = np.zeros(1, dtype=np.int64)
        self.start_days = np.asarray(self.start_days[:1], dtype=np.float32)
        self._start_rows_t = torch.as_tensor(self.start_rows, dtype=torch.int32, device=self.device)
        self._start_days_t = torch.as_tensor(self.start_days, dtype=torch.float32, device=self.device)
        self.w_off_t.zero_()

    @property
    def p(self):
        """The shared parameter block (float32, tomato_env.py:62).  ASSIGNING to it is the reference's `env.p = new_p` on a constructed env
        (experiments/run_time.py:40-41: `env.p = set_matlab_params(env.p)`): the device model, the crop block and the reward's per-step cost
        coefficients follow the new block; the reward's scale max_profit / min_profit stays what it was at construction, as in the reference
        (rewards.py:82-83 computes it once in __init__) -- glgym_set_params_keep_reward_scale.  Construct with `params=` to have both from one block."""
        return self._p

    @p.setter
    def p(self, value):
        value = np.asarray(value, dtype=np.float32)
        if value.shape != (L.NP,):
            raise ValueError(f"p: expected {L.NP} parameters, got an array of shape {value.shape}")
        p64 = np.ascontiguousarray(value, dtype=np.float64)
        L.check(self._lib.glgym_set_params_keep_reward_scale(self._h, p64.ctypes.data_as(L._DP)), "glgym_set_params_keep_reward_scale")
        self._p = value

    def _hod_table(self):
        """The reference ACCUMULATES its clock, hour_of_day = (hour_of_day + dt / 3600) % 24 once per step (tomato_env.py:127-128).  With
        dt = 900 s the increment 0.25 is exact and the sum equals timestep * dt / 3600; with dt = 300 s (experiments/run_time.py) the sum
        drifts by ulps and reads 17.999999999999996 where the product reads 18.0 -- and the rule-based controller compares the clock with
        whole hours (baseline.py:76-77, 107, 113): 13 of 960 steps of the dt = 300 hold-out switch the lamps one step apart.  The table
        holds the reference's sum for every timestep of an episode; timestep indexes it."""
        if getattr(self, "_hod_table_t", None) is None:
            t = np.empty(self.N + 4, dtype=np.float64)
            h = 0.0
            for k in range(len(t)):
                t[k] = h
                h = (h + self.dt / 3600) % 24
            self._hod_table_t = self.torch.as_tensor(t, device=self.device)
        return self._hod_table_t

    def hour_of_day(self):
        return self._hod_table()[self.timestep_t.long().clamp_(0, self.N + 3)]      # tomato_env.py:127-128, accumulated like the reference's

    def day_of_year(self):
        return self.start_day_t.double() + self.timestep_t.double() * ((self.dt / self.c) % 365)   # :126

    def rule_based_controls(self, controller):
        """u[B,6] of a gl_gym_amd.baseline.RuleBasedController for the current state (experiments/evaluate_baseline.py:22),
        computed by the glgym_rule_based kernel."""
        self._launch_rule_based(controller)
        return self.ctrl_T[:, :self.B].t().clone()

    def step_rule_based(self, controller):
        """One env-step under the rule-based controller, everything on the device (config 1, batched)."""
        return self._host_result(self.step_tensor(controller=controller))

    def metrics(self) -> Dict[str, float]:
        if self.metrics_t is None:
            return {}
        v = self.metrics_t.double().sum(dim=0)[:L.NMETRIC].cpu().numpy()      # sum of the replicas
        return {k: float(v[i]) for i, k in enumerate(L.METRIC_KEYS)}

    def capture_step_graph(self, want_obs: bool = True):
        """Capture one full step (action copy -> glgym_step -> glgym_obs -> masked glgym_reset -> masked glgym_obs) in a
        HIP graph.  The library's device-pointer entry points never synchronise, so they can be stream-captured; one
        graph launch then replaces five kernel launches (about 2 % at B = 65 536, more when B is small).  Returns
        ``replay(actions_t) -> (obs, reward, done, info)`` with the same device tensors ``step_tensor`` returns.
        Per-env crop noise (uncertainty_scale > 0) advances a host-side draw counter per step and is not capturable."""
        if self.crop_T is not None:
            raise L.GlgymError("capture_step_graph: per-step crop noise carries a host-side draw counter; use step_tensor")
        torch = self.torch
        static_a = torch.zeros(self.B, L.NU, dtype=torch.float32, device=self.device)

        def seq():
            self.action_t.copy_(static_a)
            self._action_src = self.action_t
            self._launch_step(raw_control=False)
            if want_obs:
                self._launch_obs(self.obs_t)
            if self.auto_reset:
                self._launch_reset(self.done_t)
                if want_obs:
                    self._launch_obs(self.obs_t, self.done_t, self.term_obs_t)

        # warm-up on a side stream (torch's capture protocol), with the state restored afterwards
        keep = [b.clone() for b in (self.x_T, self.u_T, self.timestep_t, self.w_off_t, self.start_day_t, self.episode_t,
                                    self.obs_t)]
        metrics = None if self.metrics_t is None else self.metrics_t.clone()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            seq()
        torch.cuda.current_stream(self.device).wait_stream(side)
        for b, k in zip((self.x_T, self.u_T, self.timestep_t, self.w_off_t, self.start_day_t, self.episode_t, self.obs_t),
                        keep):
            b.copy_(k)
        if metrics is not None:
            self.metrics_t.copy_(metrics)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            seq()
        for b, k in zip((self.x_T, self.u_T, self.timestep_t, self.w_off_t, self.start_day_t, self.episode_t, self.obs_t),
                        keep):
            b.copy_(k)                      # capture does not execute, but keep the contract explicit
        if metrics is not None:
            self.metrics_t.copy_(metrics)

        def replay(actions_t):
            static_a.copy_(actions_t.reshape(self.B, L.NU))
            graph.replay()
            return self.obs_t, self.reward_t[:self.B], self.done_t, self.info_T[:, :self.B]
        replay.graph = graph
        return replay

    def set_scheme(self, scheme: str, n_sub: Optional[int] = None, window: Optional[int] = None):
        """Switch the sub-stepper ("ls5" | "rk4" | "rk3" | "rk2", include/glgym.h), its sub-step count and tier-2b window (default: what
        this env's preset gives for the scheme; with an explicit n_sub the scheme's own window)."""
        if scheme not in L.SCHEMES:
            raise ValueError("scheme must be 'ls5', 'rk4', 'rk3' or 'rk2'")
        L.check(self._lib.glgym_set_scheme(self._h, L.SCHEMES[scheme]), "glgym_set_scheme")
        self.scheme = scheme
        n_def, w_def = L.preset_n_sub(scheme, self.dt, self.preset)
        self.set_n_sub(n_def if n_sub is None else n_sub)
        self.set_window((w_def if n_sub is None else 0) if window is None else window)

    def set_window(self, window: int):
        """Nominal sub-steps per tier-2b / harvest window (glgym_set_window): 0 = the scheme's own."""
        self.window = int(window)
        L.check(self._lib.glgym_set_window(self._h, self.window), "glgym_set_window")

    def set_layout(self, layout: str):
        """Kernel layout of float32 steps (glgym_set_layout): "auto" | "one" (lane per environment) | "quad" (four lanes)."""
        L.check(self._lib.glgym_set_layout(self._h, L.LAYOUTS[layout]), "glgym_set_layout")

    def set_occupancy(self, waves_per_simd: int):
        """0 (default): the one-lane fp32 kernel's two-waves-per-SIMD build from 131 072 environments, the one-wave build below; 1 / 2 force a build
        (include/glgym.h glgym_set_occupancy)."""
        L.check(self._lib.glgym_set_occupancy(self._h, int(waves_per_simd)), "glgym_set_occupancy")

    def set_ladder_parallel(self, on: bool):
        """Verified steps (raw controls) on up to 8 192 environments run the step-doubling ladder two rungs at a time (default; identical
        results and step_flags, two thirds of the latency: include/glgym.h glgym_set_ladder_parallel).  False: always the sequential ladder."""
        L.check(self._lib.glgym_set_ladder_parallel(self._h, int(bool(on))), "glgym_set_ladder_parallel")

    def set_n_sub(self, n_sub: int):
        self.n_sub = int(n_sub)
        L.check(self._lib.glgym_set_n_sub(self._h, self.n_sub), "glgym_set_n_sub")

    def set_verify(self, mode: str):
        """Step-doubling verified integration (include/glgym.h, glgym_verify): "auto" (default) verifies wherever the control
        can jump -- step_raw_control / controller= steps, or delta_u_max > 0.1 -- "always" / "never" every / no env-step."""
        L.check(self._lib.glgym_set_verify(self._h, L.VERIFY_MODES[mode]), "glgym_set_verify")

    def timer_start(self):
        L.check(self._lib.glgym_timer_start(self._h, self._stream()))

    def timer_stop(self) -> float:
        ms = C.c_float()
        L.check(self._lib.glgym_timer_stop(self._h, self._stream(), C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.torch.cuda.synchronize(self.device)
            self._lib.glgym_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class TomatoEnv:
    """Single-environment Gymnasium-style view over a B = 1 ``TomatoVecEnv`` (no auto-reset)."""

    def __init__(self, weather=None, params=None, dt=900.0, season_length=60, pred_horizon=0.5, dtype="float64",
                 n_sub=None, device="cuda:0", uncertainty_scale=0.0, start_day=0.0, growth_year=2010,
                 reward_params=None, constraints=None, location="synthetic", training=True, model_variant="ode",
                 scheme="rk4", observation_modules=None, u_min=None, u_max=None, delta_u_max=0.1):
        self.vec = TomatoVecEnv(1, model_variant=model_variant, scheme=scheme, observation_modules=observation_modules,
                                u_min=u_min, u_max=u_max, delta_u_max=delta_u_max,
                                weather=weather, params=params, dt=dt, season_length=season_length,
                                pred_horizon=pred_horizon, dtype=dtype, n_sub=n_sub, device=device,
                                uncertainty_scale=uncertainty_scale, start_rows=[0], start_days=[start_day],
                                reward_params=reward_params, constraints=constraints, auto_reset=False)
        v = self.vec
        self.nx, self.nu, self.nd, self.num_params, self.dt, self.c = v.nx, v.nu, v.nd, v.num_params, v.dt, v.c
        self.N, self.Np = v.N, v.Np
        self.observation_space, self.action_space = v.observation_space, v.action_space
        self.start_day, self.growth_year, self.location, self.training = start_day, growth_year, location, training
        self.u_min, self.u_max, self.delta_u_max = v.u_min, v.u_max, v.delta_u_max
        self.terminated = False

    def _set_x(self, value):          # `env.x = init_mat_state(...)` (gl_predefined_controls.py:116)
        value = np.asarray(value, dtype=np.float64).reshape(L.NX)
        self.vec.x_T[:, 0] = self.vec.torch.as_tensor(value, dtype=self.vec.tdtype, device=self.vec.device)

    x = property(lambda self: self.vec.x[0].double().cpu().numpy(), _set_x)
    weather_data = property(lambda self: self.vec.weather_data, lambda self, table: setattr(self.vec, "weather_data", table))   # run_time.py:39
    u = property(lambda self: self.vec.u[0].double().cpu().numpy())
    p = property(lambda self: self.vec.p, lambda self, value: setattr(self.vec, "p", value))     # `env.p = set_matlab_params(env.p)` (run_time.py:40)
    timestep = property(lambda self: int(self.vec.timestep_t[0]))
    day_of_year = property(lambda self: self.start_day + self.timestep * ((self.dt / self.c) % 365))
    hour_of_day = property(lambda self: float(self.vec.hour_of_day()[0]))

    def reset(self, seed: Optional[int] = None):
        obs = self.vec.reset_tensor(seed)
        self.terminated = False
        return obs[0].double().cpu().numpy(), {}

    def action_to_control(self, action):
        return np.clip(self.u + np.asarray(action) * self.delta_u_max, self.u_min, self.u_max)

    def _finish(self, out):
        obs_t, r_t, d_t, info_T = out
        info = {k: float(info_T[i, 0]) for i, k in enumerate(L.INFO_KEYS)}
        info["controls"] = self.u
        self.terminated = bool(d_t[0])
        return obs_t[0].double().cpu().numpy(), float(r_t[0]), self.terminated, False, info

    def step(self, action):
        t = self.vec.torch
        return self._finish(self.vec.step_tensor(t.as_tensor(np.asarray(action, dtype=np.float32).reshape(1, 6),
                                                             device=self.vec.device)))

    def step_raw_control(self, control):
        t = self.vec.torch
        return self._finish(self.vec.step_tensor(controls_t=t.as_tensor(np.asarray(control).reshape(1, 6),
                                                                        dtype=self.vec.tdtype, device=self.vec.device)))

    def step_raw_control_pipeinput(self, control):
        x, term = self.vec.step_raw_control_pipeinput(np.asarray(control).reshape(1, 6))
        self.terminated = bool(term[0])
        return x[0], self.terminated

    def set_crop_state(self, cBuf, cLeaf, cStem, cFruit, tCanSum):
        for i, v in zip((22, 23, 24, 25, 26), (cBuf, cLeaf, cStem, cFruit, tCanSum)):
            self.vec.x_T[i, 0] = v

    def get_obs_names(self):
        return self.vec.get_obs_names()

    def close(self):
        self.vec.close()
