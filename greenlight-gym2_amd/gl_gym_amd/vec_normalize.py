"""On-device VecNormalize around a ``TomatoVecEnv``.

Same statistics and call surface as ``stable_baselines3.common.vec_env.VecNormalize`` as the reference configures it
(gl_gym/RL/experiment_manager.py:142-147: norm_obs, norm_reward, clip_obs=10, gamma; eval envs: training=False,
norm_reward=False, gl_gym/RL/utils.py:62-66) -- but the running moments, the normalisation and the clipping run in HIP
kernels on the observation block that already lives in HBM (libglgym.so: glgym_vecnorm), so a device-resident RL loop
never copies the 263-float observations to the host.  ``obs_rms`` / ``ret_rms`` expose mean / var / count like SB3's
RunningMeanStd (used by gl_gym/common/callbacks.py:295-296), ``unnormalize_obs`` serves gl_gym/common/evaluation.py:102.
"""
from __future__ import annotations

import ctypes as C
import pickle
from types import SimpleNamespace

import numpy as np

from . import _lib as L


class VecNormalizeGPU:
    def __init__(self, venv, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0, clip_reward=10.0,
                 gamma=0.99, epsilon=1e-8):
        self.venv, self.torch = venv, venv.torch
        t, dev = venv.torch, venv.device
        self.num_envs, self.obs_dim = venv.num_envs, venv.obs_dim
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon
        f64 = dict(dtype=t.float64, device=dev)
        self.obs_mean, self.obs_var = t.zeros(self.obs_dim, **f64), t.ones(self.obs_dim, **f64)
        self.obs_count = t.full((1,), 1e-4, **f64)
        self.ret_stats = t.tensor([0.0, 1.0, 1e-4], **f64)
        self.returns = t.zeros(self.num_envs, **f64)
        self._ws = t.zeros(32 * self.obs_dim + 2, **f64)          # glgym_vecnorm_args.workspace
        self.obs_norm_t = t.zeros(self.num_envs, self.obs_dim, dtype=t.float32, device=dev)
        self.reward_norm_t = t.zeros(self.num_envs, dtype=t.float32, device=dev)
        self.old_obs = self.old_reward = None
        self._actions = None

    # ---- SB3-style views of the statistics -------------------------------------------------------
    @property
    def obs_rms(self):
        return SimpleNamespace(mean=self.obs_mean.cpu().numpy(), var=self.obs_var.cpu().numpy(),
                               count=float(self.obs_count))

    @property
    def ret_rms(self):
        s = self.ret_stats.cpu().numpy()
        return SimpleNamespace(mean=float(s[0]), var=float(s[1]), count=float(s[2]))

    def _call(self, obs_t, reward_t, done_t):
        v = self.venv
        a = L.VecNormArgs(self.num_envs, self.obs_dim, obs_t.data_ptr(), self.obs_norm_t.data_ptr(),
                          reward_t.data_ptr() if reward_t is not None else None, self.reward_norm_t.data_ptr(),
                          done_t.data_ptr() if done_t is not None else None, self.obs_mean.data_ptr(),
                          self.obs_var.data_ptr(), self.obs_count.data_ptr(), self.ret_stats.data_ptr(),
                          self.returns.data_ptr(), self._ws.data_ptr(), self.gamma, self.epsilon, self.clip_obs,
                          self.clip_reward, int(self.training), int(self.norm_obs), int(self.norm_reward))
        L.check(v._lib.glgym_vecnorm(v._h, C.byref(a), v._stream()), "glgym_vecnorm")

    # ---- tensor interface ----------------------------------------------------------------------------
    def reset_tensor(self, seed=None):
        obs = self.venv.reset_tensor(seed)
        self.returns.zero_()
        self._call(obs, None, None)
        return self.obs_norm_t

    def step_tensor(self, actions_t=None, controls_t=None):
        obs, rew, done, info = self.venv.step_tensor(actions_t, controls_t)
        self._call(obs, self.venv.reward_t, done)
        return self.obs_norm_t, self.reward_norm_t, done, info

    # ---- VecEnv calling convention ------------------------------------------------------------------------
    def reset(self):
        return self.reset_tensor().cpu().numpy()

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        t = self.torch
        base = self.venv
        while hasattr(base, "venv"):
            base = base.venv
        base._keep_applied_u = True                 # infos report the controls applied in this step
        try:
            obs_n, rew_n, done, info_T = self.step_tensor(t.as_tensor(self._actions, device=self.venv.device))
        finally:
            base._keep_applied_u = False
        # infos come from the wrapped env (so a VecMonitorGPU underneath keeps its "episode" entries); the terminal
        # observations are handed over normalised, as SB3's VecNormalize does
        term = None
        if self.venv.auto_reset and bool(done.any()):
            term = self.normalize_obs(self.venv.term_obs_t.cpu().numpy())
        dones, infos = self.venv.host_infos(done, info_T, term)
        return obs_n.cpu().numpy(), rew_n.cpu().numpy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def normalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        r = self.obs_rms
        return np.clip((obs - r.mean) / np.sqrt(r.var + self.epsilon), -self.clip_obs, self.clip_obs).astype(np.float32)

    def unnormalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        r = self.obs_rms
        return obs * np.sqrt(r.var + self.epsilon) + r.mean

    def get_original_obs(self):
        return self.venv.obs_t.cpu().numpy()

    def get_original_reward(self):
        return self.venv.reward_t[:self.num_envs].float().cpu().numpy()

    # ---- persistence: SB3's VecNormalize.save / VecNormalize.load (callbacks.py:292, experiment_manager.py:360,
    # experiments/evaluate_rl.py:31) --------------------------------------------------------------------------------
    def save(self, path):
        """Writes the statistics under SB3's attribute names (obs_rms / ret_rms with mean, var, count; clip_obs,
        clip_reward, gamma, epsilon, norm_obs, norm_reward, training) as a plain dict, so the file needs neither SB3 nor
        this package to be read.  ``load`` reads this, the round-1 layout, and files written by SB3's own
        ``VecNormalize.save`` (a pickled VecNormalize object)."""
        o, r = self.obs_rms, self.ret_rms
        with open(path, "wb") as f:
            pickle.dump(dict(format="glgym-vecnormalize-2",
                             obs_rms=dict(mean=o.mean, var=o.var, count=o.count),
                             ret_rms=dict(mean=r.mean, var=r.var, count=r.count),
                             clip_obs=self.clip_obs, clip_reward=self.clip_reward, gamma=self.gamma,
                             epsilon=self.epsilon, norm_obs=self.norm_obs, norm_reward=self.norm_reward,
                             training=self.training), f)

    @staticmethod
    def _read_stats(path):
        """-> dict(obs_mean, obs_var, obs_count, ret_mean, ret_var, ret_count, + settings found) from any of the layouts."""
        class _Stub:                                   # stands in for classes of packages that are not installed
            def __init__(self, *a, **k):
                pass

            def __setstate__(self, state):
                self.__dict__.update(state if isinstance(state, dict) else {})

        class _Unpickler(pickle.Unpickler):            # an SB3-written file references stable_baselines3 / gymnasium
            def find_class(self, module, name):        # classes; only their attribute dicts are needed here
                try:
                    return super().find_class(module, name)
                except (ImportError, AttributeError):
                    return type(name, (_Stub,), {})

        with open(path, "rb") as f:
            d = _Unpickler(f).load()

        def rms(v):
            g = (lambda k: v[k]) if isinstance(v, dict) else (lambda k: getattr(v, k))
            return np.asarray(g("mean"), dtype=np.float64), np.asarray(g("var"), dtype=np.float64), float(g("count"))

        out = {}
        if isinstance(d, dict) and "obs_mean" in d:    # round-1 layout (torch tensors)
            out.update(obs_mean=np.asarray(d["obs_mean"]), obs_var=np.asarray(d["obs_var"]),
                       obs_count=float(np.asarray(d["obs_count"]).reshape(-1)[0]))
            rs = np.asarray(d["ret_stats"], dtype=np.float64)
            out.update(ret_mean=float(rs[0]), ret_var=float(rs[1]), ret_count=float(rs[2]))
            get = d.get
        else:                                          # SB3 attribute names: our dict, or a VecNormalize object
            get = d.get if isinstance(d, dict) else (lambda k, default=None: getattr(d, k, default))
            if get("obs_rms") is None or get("ret_rms") is None:
                raise ValueError(f"{path}: neither a VecNormalizeGPU file nor an SB3 VecNormalize pickle")
            m, v, c = rms(get("obs_rms"))
            out.update(obs_mean=m, obs_var=v, obs_count=c)
            m, v, c = rms(get("ret_rms"))
            out.update(ret_mean=float(m), ret_var=float(v), ret_count=c)
        for k in ("clip_obs", "clip_reward", "gamma", "epsilon", "norm_obs", "norm_reward", "training"):
            if get(k) is not None:
                out[k] = get(k)
        return out

    def _set_stats(self, d):
        if np.shape(d["obs_mean"]) != (self.obs_dim,):
            raise ValueError(f"saved statistics are for {np.shape(d['obs_mean'])} observations, the env has {self.obs_dim}")
        t, dev = self.torch, self.venv.device
        self.obs_mean.copy_(t.as_tensor(np.asarray(d["obs_mean"], dtype=np.float64), device=dev))
        self.obs_var.copy_(t.as_tensor(np.asarray(d["obs_var"], dtype=np.float64), device=dev))
        self.obs_count.fill_(float(d["obs_count"]))
        self.ret_stats.copy_(t.tensor([d["ret_mean"], d["ret_var"], d["ret_count"]], dtype=t.float64, device=dev))

    @classmethod
    def load(cls, load_path, venv):
        """``VecNormalize.load(path, venv)`` (experiments/evaluate_rl.py:31): a wrapper around ``venv`` with the saved
        statistics and settings.  Accepts files written by ``save`` and by SB3's ``VecNormalize.save`` (e.g. the
        reference's ``best_vecnormalize.pkl``, common/callbacks.py:292) -- the latter also where SB3 is not installed."""
        d = cls._read_stats(load_path)
        kw = {k: d[k] for k in ("norm_obs", "norm_reward", "clip_obs", "clip_reward", "gamma", "epsilon", "training")
              if k in d}
        self = cls(venv, **kw)
        self._set_stats(d)
        return self

    def load_stats(self, path):
        self._set_stats(self._read_stats(path))

    def __getattr__(self, name):          # get_attr / env_method / metrics / close ... fall through to the wrapped env
        if name in ("venv", "step_async", "step_wait", "step", "reset"):      # never the inner env's (un-normalised) ones
            raise AttributeError(name)
        return getattr(self.venv, name)
