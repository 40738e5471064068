#!/usr/bin/env python3
"""One configuration of tools/window_cost.py under the profiler: B = 8, the four-lanes-per-environment kernel, ls5 n_sub 128 at the given
window, 40 uniform env-steps (python tools/window_one.py float64|float32 WINDOW).  With --pmc SQ_INSTS_VALU the difference between window 1
and window 2 is 64 windows' vector instructions at equal stage count."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))
from gl_gym_amd.tomato_env import TomatoVecEnv  # noqa: E402
from gl_gym_amd.utils import synthetic_weather  # noqa: E402
dtype, window = sys.argv[1], int(sys.argv[2])
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
env = TomatoVecEnv(8, weather=w, dtype=dtype, scheme="ls5", n_sub=128, window=window, season_length=60, pred_horizon=0.5, seed=666, start_rows=[960], auto_reset=True)
if dtype == "float32":
    env.set_layout("quad")
env.reset_tensor()
g = torch.Generator(device=env.device).manual_seed(666)
for i in range(40):
    a = torch.rand(1, 6, generator=g, device=env.device) * 2 - 1
    env.action_t.copy_(a.expand(8, 6))
    env._launch_step(raw_control=False)
    env._launch_reset(env.done_t)
torch.cuda.synchronize()
env.close()
