"""On-device episode statistics: what ``stable_baselines3.common.vec_env.VecMonitor`` adds in the reference's
``make_vec_env`` stack (gl_gym/RL/utils.py:60-61: SubprocVecEnv -> VecMonitor -> VecNormalize).

Per env the episode return (sum of RAW rewards) and length are accumulated on the device; when an env finishes, its
``infos[i]["episode"] = {"r": return, "l": length, "t": seconds since the monitor was created}`` entry is added -- the
key SB3's ``OffPolicyAlgorithm`` / ``OnPolicyAlgorithm`` episode-info buffers and the reference's evaluation code read --
and, if ``filename`` is given, a row is appended to a ``*.monitor.csv`` file in SB3's format.  Device-resident loops read
``last_returns_t`` / ``last_lengths_t`` / ``finished_t`` instead and never touch the host.

SB3 is third-party and absent here; this follows its published algorithm (pinned ==2.6.0 in the reference's
requirements.txt): VecMonitor.step_wait accumulates ``episode_returns += rewards; episode_lengths += 1`` and resets
both where ``dones``.
"""
from __future__ import annotations

import json
import time

import numpy as np


class VecMonitorGPU:
    def __init__(self, venv, filename=None, info_keywords=()):
        self.venv, self.torch = venv, venv.torch
        t, dev = venv.torch, venv.device
        self.num_envs = venv.num_envs
        self.observation_space, self.action_space = venv.observation_space, venv.action_space
        self.episode_returns = t.zeros(self.num_envs, dtype=t.float64, device=dev)
        self.episode_lengths = t.zeros(self.num_envs, dtype=t.int32, device=dev)
        self.last_returns_t = t.zeros(self.num_envs, dtype=t.float64, device=dev)   # of the episode that just ended
        self.last_lengths_t = t.zeros(self.num_envs, dtype=t.int32, device=dev)
        self.finished_t = t.zeros(self.num_envs, dtype=t.uint8, device=dev)
        self.episode_count = 0
        self.t_start = time.time()
        self.info_keywords = tuple(info_keywords)
        self._file = None
        self._actions = None
        if filename is not None:
            if not filename.endswith("monitor.csv"):
                filename = filename + ".monitor.csv"
            self._file = open(filename, "wt")
            self._file.write("#%s\n" % json.dumps({"t_start": self.t_start, "env_id": "TomatoEnv"}))
            self._file.write(",".join(("r", "l", "t") + self.info_keywords) + "\n")
            self._file.flush()

    # ---- tensor interface ----------------------------------------------------------------------------
    def reset_tensor(self, seed=None):
        obs = self.venv.reset_tensor(seed)
        self.episode_returns.zero_()
        self.episode_lengths.zero_()
        return obs

    def step_tensor(self, actions_t=None, controls_t=None, want_obs=True, controller=None):
        out = self.venv.step_tensor(actions_t, controls_t, want_obs, controller)
        _, r_t, d_t, _ = out
        self.episode_returns += r_t[:self.num_envs].double()
        self.episode_lengths += 1
        done = d_t.bool()
        self.finished_t.copy_(d_t)
        self.last_returns_t.copy_(self.torch.where(done, self.episode_returns, self.last_returns_t))
        self.last_lengths_t.copy_(self.torch.where(done, self.episode_lengths, self.last_lengths_t))
        self.episode_returns.masked_fill_(done, 0.0)
        self.episode_lengths.masked_fill_(done, 0)
        return out

    # ---- VecEnv calling convention ----------------------------------------------------------------------
    def reset(self):
        return self.venv._obs_to_host(self.reset_tensor())

    def host_infos(self, d_t, info_T, term_obs=None):
        dones, infos = self.venv.host_infos(d_t, info_T, term_obs)
        if dones.any():
            idx = np.nonzero(dones)[0]
            rets = self.last_returns_t.cpu().numpy()
            lens = self.last_lengths_t.cpu().numpy()
            now = round(time.time() - self.t_start, 6)
            for b in idx:
                d = infos[b]
                ep = {"r": float(rets[b]), "l": int(lens[b]), "t": now}
                for k in self.info_keywords:
                    ep[k] = d[k]
                d["episode"] = ep
                infos[b] = d
                if self._file is not None:
                    self._file.write(",".join(str(ep[k]) for k in ("r", "l", "t") + self.info_keywords) + "\n")
            self.episode_count += len(idx)
            if self._file is not None:
                self._file.flush()
        return dones, infos

    def step_async(self, actions):
        self._actions = np.asarray(actions, dtype=np.float32)

    def step_wait(self):
        t = self.torch
        base = self.venv
        while hasattr(base, "venv"):                # the env at the bottom of the wrapper stack owns the flag
            base = base.venv
        base._keep_applied_u = True                 # infos report the controls applied in this step
        try:
            obs_t, r_t, d_t, info_T = self.step_tensor(t.as_tensor(self._actions, device=self.venv.device))
        finally:
            base._keep_applied_u = False
        dones, infos = self.host_infos(d_t, info_T)
        return self.venv._obs_to_host(obs_t), r_t.float().cpu().numpy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if self._file is not None:
            self._file.close()
            self._file = None
        self.venv.close()

    def __getattr__(self, name):          # everything else (get_attr, obs_t, reward_t, metrics ...) is the wrapped env's
        if name in ("venv", "step_async", "step_wait", "step", "reset"):      # never the inner env's (unmonitored) ones
            raise AttributeError(name)
        return getattr(self.venv, name)
