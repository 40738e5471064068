#!/usr/bin/env python3
"""Diagnostic (GPU): mean step-kernel time of the bench workload (B = 65 536, fp32) against the nominal n_sub -- the nominal sub-step must
cover the rate bound of practically every lane (a launch lasts as long as its slowest lane), but every lane pays for it.
    python tools/nsub_sweep.py [scheme] [steps]"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
scheme = sys.argv[1] if len(sys.argv) > 1 else "ls5"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
B = 65536
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024); starts = np.arange(0, 35040 - 5760 - 60, 96)
for n_sub in ((104, 112, 120, 124, 128, 132, 136, 144) if scheme == "ls5" else (224, 232, 240, 248, 256)):
    env = TomatoVecEnv(B, weather=w, dtype="float32", scheme=scheme, n_sub=n_sub, season_length=60, pred_horizon=0.5, seed=666, start_rows=starts, auto_reset=True)
    env.reset_tensor()
    env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=env.device, generator=torch.Generator(device=env.device).manual_seed(1234)).to(env.tdtype))
    g = torch.Generator(device=env.device).manual_seed(666)
    ms, mx = [], []
    for i in range(steps):
        env.action_t.uniform_(-1.0, 1.0, generator=g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env._launch_step(raw_control=False); e1.record()
        env._launch_reset(env.done_t)
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1)); mx.append(int(((env.step_flags_t >> 16) & 0x7fff).max()))
    ms, mx = np.array(ms[40:]), np.array(mx[40:])
    m = env.metrics()
    print(f"{scheme} n_sub {n_sub}: kernel ms mean {ms.mean():.4f} median {np.median(ms):.4f} min {ms.min():.4f}; slowest lane's extra sub-steps per launch: median {np.median(mx):.0f} mean {mx.mean():.1f}; "
          f"retries {m.get('n_guard_retries')}, failed {m.get('n_ode_fail')}", flush=True)
    env.close()
