"""Diagnostic: run the bench workload and save the (x, u, d) tuples of env-steps the kernel flagged as failed integrations
or refined heavily, for offline analysis against the oracle.  Writes gpurun_out/failures.npz."""
import sys
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np, torch
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
B = 65536
w = synthetic_weather(35040); starts = np.arange(0, 35040 - 5760 - 60, 96)
env = TomatoVecEnv(B, weather=w, dtype="float32", scheme="rk4", season_length=60, start_rows=starts.tolist(),
                   start_days=(starts / 96.0).tolist(), seed=666)
env.reset_tensor()
g = torch.Generator(device=env.device).manual_seed(1)
X, U, D, K = [], [], [], []
for i in range(steps):
    a = torch.rand(B, 6, generator=g, device=env.device) * 2 - 1
    x_prev = env.x_T.clone(); u_prev = env.u_T.clone(); ts = env.timestep_t.clone(); off = env.w_off_t.clone()
    env.action_t.copy_(a)
    env._launch_step(raw_control=False)
    bad = (env.done_t != 0) & (ts < env.N)
    if bad.any():
        idx = torch.nonzero(bad).flatten()
        X.append(x_prev[:, idx].t().double().cpu().numpy())
        U.append(env.u_T[:, idx].t().double().cpu().numpy())           # applied control
        D.append(env.weather_t[(off[idx] + ts[idx]).long()].double().cpu().numpy())
        K.append(np.full(len(idx), i))
    env._launch_reset(env.done_t)
print(env.metrics())
if X:
    np.savez("gpurun_out/failures.npz", X=np.concatenate(X), U=np.concatenate(U), D=np.concatenate(D), step=np.concatenate(K))
    print("saved", sum(len(x) for x in X), "failed tuples")
