"""Throughput of the host-buffer (SB3 numpy) path vs the device-tensor path, for DESIGN.md (never bench.py's `value`)."""
import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np, torch
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
w = synthetic_weather(n_rows=35040)
for B in (8, 1024, 65536):
    env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=60, start_rows=list(range(0, 20000, 96)), seed=1)
    env.reset()
    a = np.random.default_rng(0).uniform(-1, 1, (B, 6)).astype(np.float32)
    at = torch.as_tensor(a, device=env.device)
    for _ in range(2): env.step(a)
    torch.cuda.synchronize(); t = time.perf_counter(); n = 5
    for _ in range(n): env.step(a)
    t_host = (time.perf_counter() - t) / n
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): env.step_tensor(at)
    torch.cuda.synchronize(); t_dev = (time.perf_counter() - t) / n
    # obs + reward + done over PCIe, without building the per-env info dicts
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        o, r, d, i = env.step_tensor(at); env._obs_to_host(o); r.cpu(); d.cpu()
    t_pcie = (time.perf_counter() - t) / n
    print(f"B={B:6d}  step() numpy+infos {B/t_host:10.3e} env-steps/s ({1e3*t_host:8.2f} ms) | obs/reward/done D2H only "
          f"{B/t_pcie:10.3e} ({1e3*t_pcie:7.2f} ms) | step_tensor {B/t_dev:10.3e} ({1e3*t_dev:6.2f} ms)")
    env.close()
