#!/usr/bin/env python3
"""Diagnostic (GPU): per launch of the bench workload, how many sub-steps the SLOWEST environment took (at one wave per SIMD the
launch lasts as long as its slowest lane), next to the kernel time; and the same kernel on a uniform calm batch (every lane nominal).
    python tools/substep_hist.py [steps] [n_sub] [scheme]"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_sub = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else None
scheme = sys.argv[3] if len(sys.argv) > 3 else "ls5"
import os
B = int(os.environ.get("GLGYM_TOOL_B", "65536"))
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024); starts = np.arange(0, 35040 - 5760 - 60, 96)
for label, sr in (("bench workload", starts), ("uniform batch (one start row, same actions)", [960])):
    env = TomatoVecEnv(B, weather=w, dtype="float32", scheme=scheme, n_sub=n_sub, season_length=60, pred_horizon=0.5, seed=666, start_rows=sr, auto_reset=True)
    env.reset_tensor()
    uniform = len(sr) == 1
    if not uniform:
        env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=env.device, generator=torch.Generator(device=env.device).manual_seed(1234)).to(env.tdtype))
    g = torch.Generator(device=env.device).manual_seed(666)
    mx, ms, mean = [], [], []
    for i in range(steps):
        a = torch.rand(1 if uniform else B, 6, generator=g, device=env.device) * 2 - 1
        env.action_t.copy_(a.expand(B, 6))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); env._launch_step(raw_control=False); e1.record()
        env._launch_reset(env.done_t)
        torch.cuda.synchronize()
        ex = (env.step_flags_t >> 16) & 0x7fff
        mx.append(int(ex.max())); mean.append(float(ex.float().mean())); ms.append(e0.elapsed_time(e1))
    mx, ms = np.array(mx[20:]), np.array(ms[20:])
    # per-wave maxima of the last launch: how many waves carry extra sub-steps
    wv = ex.view(-1, 64).max(dim=1).values.cpu().numpy()
    hist = np.bincount(np.minimum(ex.cpu().numpy(), 64), minlength=65)
    print(f"{label}: {scheme} n_sub {env.n_sub}; kernel ms mean {ms.mean():.3f} min {ms.min():.3f} max {ms.max():.3f}; slowest lane's extra sub-steps per launch: "
          f"mean {mx.mean():.1f} median {np.median(mx):.0f} max {mx.max()}; mean extra per env-step {np.mean(mean[20:]):.3f}; corr(ms, max) {np.corrcoef(ms, mx)[0,1]:.2f}; "
          f"last launch: waves with extra > 0: {(wv > 0).sum()} of {len(wv)}, > 8: {(wv > 8).sum()}, > 32: {(wv > 32).sum()}")
    print("   last launch, extra sub-steps per env-step histogram (0, 1-2, 3-4, 5-8, 9-16, 17-32, 33-63, 64+):",
          [int(hist[0]), int(hist[1:3].sum()), int(hist[3:5].sum()), int(hist[5:9].sum()), int(hist[9:17].sum()), int(hist[17:33].sum()), int(hist[33:64].sum()), int(hist[64])])
    k = np.argsort(mx)
    print("   (max extra, ms) at quantiles:", [(int(mx[k[int(q * (len(k) - 1))]]), round(float(ms[k[int(q * (len(k) - 1))]]), 3)) for q in (0, .25, .5, .75, 1)])
    env.close()
