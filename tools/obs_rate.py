"""Time glgym_obs alone (B = 65 536, Np = 48) -- used to tune the kernel's launch shape."""
import sys
sys.path.insert(0, "greenlight-gym2_amd")
import torch
from gl_gym_amd.tomato_env import TomatoVecEnv
env = TomatoVecEnv(65536, dtype="float32", season_length=10, auto_reset=False)
env.reset_tensor()
for _ in range(5): env._launch_obs(env.obs_t)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(100): env._launch_obs(env.obs_t)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 10
print(f"obs_kernel: {us:.1f} us per call, {65536 * env.obs_dim * 4 / us / 1e6:.2f} TB/s written")
