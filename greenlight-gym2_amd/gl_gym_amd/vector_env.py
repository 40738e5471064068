"""``gymnasium.vector.VectorEnv`` facade over ``TomatoVecEnv`` (north_star: "keeping the Gymnasium Env/VectorEnv and
reset()/step() API"; the reference's env base class is gymnasium's, gl_gym/environments/base_env.py:14,173).

    venv = TomatoVectorEnv(num_envs=4096, weather=...)          # same keyword arguments as TomatoVecEnv
    obs, infos = venv.reset(seed=666)                            # obs [B, 263] float32, infos {}
    obs, rewards, terminations, truncations, infos = venv.step(actions)        # actions [B, 6] in [-1, 1]

Semantics (Gymnasium's vector API):
  * five-tuple ``step``; ``truncations`` is all False (the reference never truncates: tomato_env.py:141-146 returns False);
  * ``infos`` is a dict of arrays: one [B] float64 array per key of tomato_env.py:208-222 (``EPI`` ... ``lamp_violation``),
    ``controls`` [B, 6], and a boolean ``_<key>`` mask per key (all True: every env reports every step);
  * autoreset mode SAME_STEP (``metadata["autoreset_mode"]``; gymnasium >= 1.0 ``AutoresetMode.SAME_STEP``, the behaviour
    of gymnasium 0.29's vector envs and of SB3): an env that terminates is reset inside the same ``step`` call, the
    returned observation row is the first observation of its NEW episode, and the last observation / info of the finished
    episode are in ``infos["final_obs"]`` (gymnasium >= 1.0 name) and ``infos["final_observation"]`` (0.29 name), both
    object arrays with ``None`` for running envs, and ``infos["final_info"]`` (a dict of arrays for the finished envs'
    last step = the values of this very step), with their ``_final_*`` masks = ``terminations``;
  * ``reset(seed=s)`` re-seeds the episode-start draws (env b uses stream (s, b)); ``options`` is accepted and ignored.

When ``gymnasium`` is importable the class derives from ``gymnasium.vector.VectorEnv`` and its spaces are gymnasium Boxes
(``batch_space``); otherwise it is a plain class with the same attributes.  The device-tensor interface of the wrapped env
(``step_tensor`` / ``reset_tensor``) stays reachable through ``.venv``.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import numpy as np

from . import _lib as L
from .tomato_env import TomatoVecEnv, _box

try:  # pragma: no cover - depends on the installation
    import gymnasium as _gym
    _Base = _gym.vector.VectorEnv
except Exception:  # gymnasium absent: same surface, no base class
    _gym = None
    _Base = object


class TomatoVectorEnv(_Base):
    metadata = {"autoreset_mode": "same_step", "render_modes": []}

    def __init__(self, num_envs: int, **kwargs):
        kwargs.setdefault("auto_reset", True)
        if not kwargs["auto_reset"]:
            raise ValueError("the VectorEnv facade autoresets (same-step); use TomatoVecEnv for manual resets")
        kwargs.setdefault("lazy_infos", False)
        self.venv = TomatoVecEnv(num_envs, **kwargs)
        self.num_envs = self.venv.num_envs
        self.single_observation_space = self.venv.observation_space
        self.single_action_space = self.venv.action_space
        lo, hi = self.single_observation_space.low, self.single_observation_space.high
        if _gym is not None:  # pragma: no cover
            self.single_observation_space = _gym.spaces.Box(lo, hi, dtype=np.float32)
            self.single_action_space = _gym.spaces.Box(-1.0, 1.0, (L.NU,), dtype=np.float32)
            self.observation_space = _gym.vector.utils.batch_space(self.single_observation_space, self.num_envs)
            self.action_space = _gym.vector.utils.batch_space(self.single_action_space, self.num_envs)
        else:
            self.observation_space = _box(np.tile(lo, (self.num_envs, 1)), np.tile(hi, (self.num_envs, 1)),
                                          (self.num_envs, self.venv.obs_dim), np.float32)
            self.action_space = _box(-1.0, 1.0, (self.num_envs, L.NU), np.float32)
        self.closed = False

    # ---- Gymnasium vector API ---------------------------------------------------------------------
    def reset(self, *, seed: Optional[int] = None, options: Optional[Dict[str, Any]] = None) -> Tuple[np.ndarray, Dict]:
        obs_t = self.venv.reset_tensor(seed)
        return np.array(self.venv._obs_to_host(obs_t)), {}

    def step(self, actions):
        v = self.venv
        actions = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, L.NU)
        v._keep_applied_u = True
        try:
            obs_t, r_t, d_t, info_T = v.step_tensor(v.torch.as_tensor(actions, device=v.device))
        finally:
            v._keep_applied_u = False
        obs = np.array(v._obs_to_host(obs_t))
        rewards = r_t.double().cpu().numpy()
        term = d_t.cpu().numpy().astype(bool)
        rows = info_T.double().cpu().numpy()                         # [11, B]
        applied = v._u_applied_T if getattr(v, "_u_applied_valid", False) else v.u_T
        infos: Dict[str, Any] = {}
        all_true = np.ones(self.num_envs, dtype=bool)
        for i, key in enumerate(L.INFO_KEYS):
            infos[key] = rows[i].copy()
            infos["_" + key] = all_true
        infos["controls"] = applied[:, :self.num_envs].t().double().cpu().numpy()
        infos["_controls"] = all_true
        if term.any():
            final_obs = np.full(self.num_envs, None, dtype=object)
            t_obs = v.term_obs_t.cpu().numpy()
            for b in np.nonzero(term)[0]:
                final_obs[b] = t_obs[b].copy()
            final_info = {key: np.where(term, infos[key], 0.0) for key in L.INFO_KEYS}
            final_info.update({"_" + key: term.copy() for key in L.INFO_KEYS})
            infos["final_obs"] = infos["final_observation"] = final_obs
            infos["_final_obs"] = infos["_final_observation"] = term.copy()
            infos["final_info"], infos["_final_info"] = final_info, term.copy()
        return obs, rewards, term, np.zeros(self.num_envs, dtype=bool), infos

    def close(self, **kwargs):
        if not self.closed:
            self.venv.close()
            self.closed = True

    def close_extras(self, **kwargs):  # gymnasium's VectorEnv.close() hook
        self.venv.close()

    # ---- what the reference's consumers reach through the env --------------------------------
    def get_attr(self, name: str):
        return tuple(self.venv.get_attr(name))

    def call(self, name: str, *args, **kwargs):
        return tuple(self.venv.env_method(name, *args, **kwargs))

    def set_attr(self, name: str, values):
        self.venv.set_attr(name, values)

    @property
    def unwrapped(self):
        return self

    def __getattr__(self, name):
        if name in ("venv",):
            raise AttributeError(name)
        return getattr(self.venv, name)
