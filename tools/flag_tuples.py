#!/usr/bin/env python3
"""Diagnostic (GPU): run the bench workload and save the (x, u, d) tuples of every env-step whose first integration attempt was
not accepted as it stood (step_flags, include/glgym.h GLGYM_SF_*), for offline analysis against the CPU checker.
    python tools/flag_tuples.py [steps] [n_sub] [dtype] [scheme]      -> gpurun_out/flagged_tuples.npz
GLGYM_TOOL_HEAVY=k: also the env-steps that took at least k sub-steps beyond n_sub (the launch waits for its slowest lane)."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
scheme = sys.argv[4] if len(sys.argv) > 4 else "ls5"
dtype = sys.argv[3] if len(sys.argv) > 3 else "float32"
import os
B = int(os.environ.get("GLGYM_TOOL_B", "65536"))
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024); starts = np.arange(0, 35040 - 5760 - 60, 96)
env = TomatoVecEnv(B, weather=w, dtype=dtype, scheme=scheme, n_sub=n_sub, season_length=60, pred_horizon=0.5, seed=666, start_rows=starts,
                   auto_reset=True)
env.reset_tensor()
env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=env.device, generator=torch.Generator(device=env.device).manual_seed(1234)).to(env.tdtype))
g = torch.Generator(device=env.device).manual_seed(666)
X, U, D, F, K = [], [], [], [], []
hist = {}
for i in range(steps):
    a = torch.rand(B, 6, generator=g, device=env.device) * 2 - 1
    x_prev = env.x_T.clone(); ts = env.timestep_t.clone(); off = env.w_off_t.clone()
    env.action_t.copy_(a)
    env._launch_step(raw_control=False)
    fl = env.step_flags_t
    heavy_min = int(os.environ.get("GLGYM_TOOL_HEAVY", "0"))   # > 0: also capture env-steps with at least that many sub-steps beyond n_sub
    sel = (fl & 0xffff) != 0                                   # first-attempt flags / extra attempts (bits 16.. only count sub-steps)
    if heavy_min > 0:
        sel = sel | ((fl >> 16) >= heavy_min)
    idx = torch.nonzero(sel).flatten()
    if len(idx):
        X.append(x_prev[:, idx].t().double().cpu().numpy())
        U.append(env.u_T[:, idx].t().double().cpu().numpy())           # applied control
        D.append(env.weather_t[(off[idx] + ts[idx]).long()].double().cpu().numpy())
        F.append(fl[idx].cpu().numpy()); K.append(np.full(len(idx), i))
        for f in F[-1]: hist[int(f) & 0xffff] = hist.get(int(f) & 0xffff, 0) + 1
    env._launch_reset(env.done_t)
print(scheme, "n_sub", env.n_sub, dtype, "flag words seen (word: count):", dict(sorted(hist.items())), "of", steps * B, "env-steps")
print(env.metrics())
if X:
    out = ROOT / "gpurun_out" / "flagged_tuples.npz"
    out.parent.mkdir(exist_ok=True)
    np.savez(out, X=np.concatenate(X), U=np.concatenate(U), D=np.concatenate(D), flags=np.concatenate(F), step=np.concatenate(K))
    print("saved", sum(len(x) for x in X), "tuples to", out)
