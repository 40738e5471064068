#!/usr/bin/env python3
"""Device-resident rollout collection: TomatoVecEnv -> VecMonitorGPU -> VecNormalizeGPU -> a torch policy, with the
observation / action / reward tensors never leaving HBM ("zero-copy hand-off to a torch-native rollout
buffer").  It is the data path of the reference's PPO loop (stable_baselines3 OnPolicyAlgorithm.collect_rollouts over
make_vec_env's stack, gl_gym/RL/experiment_manager.py:95-147) with SB3's numpy round trip removed; the optimiser step
is out of scope here.

    python examples/device_rollout.py --n-envs 65536 --n-steps 64
"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))

import torch  # noqa: E402

from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
from gl_gym_amd.vec_monitor import VecMonitorGPU  # noqa: E402
from gl_gym_amd.vec_normalize import VecNormalizeGPU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=65536)
    ap.add_argument("--n-steps", type=int, default=64, help="rollout length per env (PPO n_steps)")
    ap.add_argument("--scheme", default="ls5", choices=["ls5", "rk4", "rk3", "rk2"])
    ap.add_argument("--hidden", type=int, default=64)
    args = ap.parse_args()

    w = synthetic_weather(n_rows=35040)
    starts = list(range(0, 35040 - 5760 - 60, 96))
    base = TomatoVecEnv(args.n_envs, weather=w, dtype="float32", scheme=args.scheme, season_length=60, start_rows=starts,
                        start_days=[s / 96.0 for s in starts], seed=666)
    env = VecNormalizeGPU(VecMonitorGPU(base), norm_obs=True, norm_reward=True, clip_obs=10.0, gamma=0.9631)
    dev, B, T, D = base.device, args.n_envs, args.n_steps, base.obs_dim

    torch.manual_seed(0)
    pi = torch.nn.Sequential(torch.nn.Linear(D, args.hidden), torch.nn.Tanh(), torch.nn.Linear(args.hidden, args.hidden),
                             torch.nn.Tanh(), torch.nn.Linear(args.hidden, 6)).to(dev)
    vf = torch.nn.Sequential(torch.nn.Linear(D, args.hidden), torch.nn.Tanh(), torch.nn.Linear(args.hidden, 1)).to(dev)
    log_std = torch.zeros(6, device=dev)

    # rollout buffer on the device (SB3 RolloutBuffer fields)
    buf = dict(obs=torch.empty(T, B, D, device=dev), act=torch.empty(T, B, 6, device=dev), rew=torch.empty(T, B, device=dev),
               done=torch.empty(T, B, device=dev), val=torch.empty(T, B, device=dev), logp=torch.empty(T, B, device=dev))
    obs = env.reset_tensor()

    def collect():
        nonlocal obs
        with torch.no_grad():
            for t in range(T):
                mean, value = pi(obs), vf(obs).squeeze(-1)
                noise = torch.randn_like(mean)
                act = mean + noise * log_std.exp()
                buf["obs"][t].copy_(obs); buf["act"][t].copy_(act); buf["val"][t].copy_(value)
                buf["logp"][t].copy_((-0.5 * noise.pow(2) - log_std - 0.9189385).sum(-1))
                obs, rew, done, _ = env.step_tensor(act.clamp(-1.0, 1.0))
                buf["rew"][t].copy_(rew); buf["done"][t].copy_(done)
            # GAE(lambda) backwards over the buffer, still on the device
            gamma, lam = 0.9631, 0.95
            adv = torch.zeros(B, device=dev); last_v = vf(obs).squeeze(-1)
            advs = torch.empty(T, B, device=dev)
            for t in reversed(range(T)):
                nonterm = 1.0 - buf["done"][t]
                delta = buf["rew"][t] + gamma * last_v * nonterm - buf["val"][t]
                adv = delta + gamma * lam * nonterm * adv
                advs[t] = adv; last_v = buf["val"][t]
        return advs

    collect()                                            # warm-up (allocator, kernel code objects)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    advs = collect()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    mon = env.venv
    print(f"{B} envs x {T} steps collected in {el * 1e3:.1f} ms: {B * T / el:.3e} env-steps/s end to end "
          f"(policy + value nets, env step, monitor, VecNormalize, GAE; scheme {args.scheme})")
    print(f"advantage mean {float(advs.mean()):+.4f}, normalised reward mean {float(buf['rew'].mean()):+.4f}, "
          f"episodes finished so far {int(mon.finished_t.sum())}, ODE failures {base.metrics()['n_ode_fail']:.0f}")
    base.close()


if __name__ == "__main__":
    main()
