#!/usr/bin/env python3
"""Env-steps per second of the fused step at small batch sizes, one lane per environment against four lanes per environment
(GLGYM_LAYOUT; include/glgym.h).   python tools/small_batch_rate.py [float32|float64]"""
import os, subprocess, sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
if len(sys.argv) > 2:                                   # child: one layout
    sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
    import numpy as np, torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    dtype = sys.argv[1]
    w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
    sizes = [int(v) for v in os.environ["GLGYM_RATE_SIZES"].split(",")] if os.environ.get("GLGYM_RATE_SIZES") else (8, 64, 1024, 4096, 16384, 65536)
    for B in sizes:
        env = TomatoVecEnv(B, weather=w, dtype=dtype, season_length=60, pred_horizon=0.5, seed=1, start_rows=np.arange(0, 20000, 96),
                           auto_reset=True)
        env.reset_tensor()
        gen = torch.Generator(device=env.device).manual_seed(3)
        acts = [torch.rand(B, 6, generator=gen, device=env.device) * 2 - 1 for _ in range(16)]
        n = 60 if dtype == "float32" else 12
        for i in range(n // 3):
            env.step_tensor(acts[i % 16], want_obs=False)
        env.metrics_t.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            env.step_tensor(acts[i % 16], want_obs=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        m = env.metrics()
        print(f"  {sys.argv[2]:4s} {dtype} B = {B:6d}: {1e3 * dt:7.3f} ms per step, {B / dt:.3e} env-steps/s; failed {m['n_ode_fail']:.0f}, "
              f"extra attempts {m['n_guard_retries']:.0f}")
        env.close()
else:
    dtype = sys.argv[1] if len(sys.argv) > 1 else "float32"
    for layout in ("one", "quad"):
        subprocess.run([sys.executable, __file__, dtype, layout], env=dict(os.environ, GLGYM_LAYOUT=layout))
