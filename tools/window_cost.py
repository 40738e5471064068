#!/usr/bin/env python3
"""Diagnostic (GPU): what one tier-2b window costs next to one stage.  The fused step kernel on a UNIFORM batch (every lane nominal,
B = 65 536, fp32) at fixed n_sub with 1, 2, 3, 4, 6 sub-steps per window (glgym_set_window): kernel ms = a x stages + b x windows.
    python tools/window_cost.py [B] [float32|float64] [auto|one|quad]      (round 6: any dtype / layout, e.g. `8 float64` = the four-lanes-per-
environment fp64 kernel at the reference's n_envs)"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
DTYPE = sys.argv[2] if len(sys.argv) > 2 else "float32"
LAYOUT = sys.argv[3] if len(sys.argv) > 3 else "auto"
SCHEMES = (("ls5", 128, 5), ("rk4", 240, 4), ("rk3", 270, 3)) if DTYPE == "float32" else (("ls5", 128, 5),)
print(f"# B = {B}, {DTYPE}, layout {LAYOUT}")
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
for scheme, n_sub, stages in SCHEMES:
    rows = []
    for window in (1, 2, 3, 4, 6):
        env = TomatoVecEnv(B, weather=w, dtype=DTYPE, scheme=scheme, n_sub=n_sub, window=window, season_length=60, pred_horizon=0.5, seed=666, start_rows=[960], auto_reset=True)
        if DTYPE == "float32":
            env.set_layout(LAYOUT)
        env.reset_tensor()
        g = torch.Generator(device=env.device).manual_seed(666)
        ms = []
        for i in range(120):
            a = torch.rand(1, 6, generator=g, device=env.device) * 2 - 1
            env.action_t.copy_(a.expand(B, 6))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env._launch_step(raw_control=False); e1.record()
            env._launch_reset(env.done_t)
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        t = float(np.median(ms[20:]))
        rows.append((n_sub * stages, n_sub // window, t))
        print(f"{scheme} n_sub {n_sub} window {window}: {n_sub * stages} stages, {n_sub // window} windows: kernel {t:.4f} ms (median of 100)", flush=True)
        env.close()
    A = np.array([[r[0], r[1], 1.0] for r in rows]); y = np.array([r[2] for r in rows])
    (a, b, c), *_ = np.linalg.lstsq(A, y, rcond=None)
    print(f"   fit: {a*1e3:.4f} us per stage, {b*1e3:.3f} us per window ({b/a:.2f} stages), {c*1e3:.1f} us fixed")
