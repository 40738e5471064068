"""CPU oracle for the on-device VecNormalize -- TEST INFRASTRUCTURE ONLY.

numpy restatement of stable_baselines3.common.running_mean_std.RunningMeanStd and
stable_baselines3.common.vec_env.vec_normalize.VecNormalize (step_wait / reset / normalize_* / unnormalize_obs) as
published in Stable-Baselines3 2.6.0 -- the version the reference pins (requirements.txt: stable_baselines3==2.6.0)
and configures at gl_gym/RL/experiment_manager.py:142-147 (norm_obs, norm_reward, clip_obs=10, gamma) and
gl_gym/RL/utils.py:62-66.  SB3 is a third-party dependency absent from /root/reference and from this image, so this
oracle is "parity unpinned": it follows the published algorithm, anchored on the reference's call sites only.
"""
import numpy as np


class RunningMeanStd:
    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr):
        self.update_from_moments(np.mean(arr, axis=0), np.var(arr, axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        tot_count = self.count + batch_count
        new_mean = self.mean + delta * batch_count / tot_count
        m_2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / tot_count
        self.mean, self.var, self.count = new_mean, m_2 / tot_count, tot_count


class VecNormalizeOracle:
    def __init__(self, num_envs, obs_dim, training=True, norm_obs=True, norm_reward=True, clip_obs=10.0,
                 clip_reward=10.0, gamma=0.99, epsilon=1e-8):
        self.obs_rms, self.ret_rms = RunningMeanStd(shape=(obs_dim,)), RunningMeanStd(shape=())
        self.returns = np.zeros(num_envs)
        self.training, self.norm_obs, self.norm_reward = training, norm_obs, norm_reward
        self.clip_obs, self.clip_reward, self.gamma, self.epsilon = clip_obs, clip_reward, gamma, epsilon

    def normalize_obs(self, obs):
        if not self.norm_obs:
            return obs
        return np.clip((obs - self.obs_rms.mean) / np.sqrt(self.obs_rms.var + self.epsilon), -self.clip_obs,
                       self.clip_obs).astype(np.float32)

    def unnormalize_obs(self, obs):
        return obs * np.sqrt(self.obs_rms.var + self.epsilon) + self.obs_rms.mean if self.norm_obs else obs

    def normalize_reward(self, r):
        if not self.norm_reward:
            return r
        return np.clip(r / np.sqrt(self.ret_rms.var + self.epsilon), -self.clip_reward, self.clip_reward)

    def reset(self, obs):
        self.returns = np.zeros_like(self.returns)
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        return self.normalize_obs(obs)

    def step(self, obs, rewards, dones):
        if self.training and self.norm_obs:
            self.obs_rms.update(obs)
        obs_n = self.normalize_obs(obs)
        if self.training:
            self.returns = self.returns * self.gamma + rewards
            self.ret_rms.update(self.returns)
        rew_n = self.normalize_reward(rewards)
        self.returns[dones] = 0
        return obs_n, rew_n
