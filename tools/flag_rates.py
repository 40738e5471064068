#!/usr/bin/env python3
"""How often the guard of rk4_delta_guarded acts on the bench workload, by cause and by phase of the run (GPU).
    python tools/flag_rates.py [scheme]       -> per phase: env-steps, ms per step, extra attempts, flags by cause, failed"""
import sys, time
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402

scheme = sys.argv[1] if len(sys.argv) > 1 else "rk4"
B = 65536
weather = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
starts = np.arange(0, 35040 - 5760 - 60, 96)
env = TomatoVecEnv(B, weather=weather, dtype="float32", scheme=scheme, season_length=60, pred_horizon=0.5, seed=666,
                   start_rows=starts, auto_reset=True)
env.reset_tensor()
dev = env.device
gen = torch.Generator(device=dev).manual_seed(666)
done = 0
for (n, label) in ((8, "steps 0-8 (from the reset state: every exchange law on its kink)"), (42, "steps 8-50"), (200, "steps 50-250"),
                   (1000, "steps 250-1250"), (1000, "steps 1250-2250")):
    env.metrics_t.zero_()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        env.action_t.copy_(torch.rand(B, 6, generator=gen, device=dev) * 2 - 1)
        env._launch_step(raw_control=False)
        env._launch_obs(env.obs_t)
        env._launch_reset(env.done_t)
        env._launch_obs(env.obs_t, env.done_t, env.term_obs_t)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    m = env.metrics()
    print(f"{scheme} {label}: {m['n_env_steps']:.3g} env-steps, {1e3 * dt / n:.3f} ms/step; extra attempts {m['n_guard_retries']:.0f}, "
          f"refined sub-steps {m['n_refined_substeps']:.0f}; first-attempt flags: branch {m['n_flag_branch']:.0f} err {m['n_flag_err']:.0f} "
          f"cap/nonfinite {m['n_flag_cap']:.0f} heavy {m['n_flag_heavy']:.0f}; failed {m['n_ode_fail']:.0f}")
