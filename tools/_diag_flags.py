import sys, numpy as np, torch
sys.path.insert(0, "greenlight-gym2_amd")
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
w = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
B = 65536
env = TomatoVecEnv(B, weather=w, dtype="float32", season_length=60, pred_horizon=0.5, seed=1, start_rows=np.arange(0, 20000, 96), auto_reset=True)
env.reset_tensor()
gen = torch.Generator(device=env.device).manual_seed(3)
acts = [torch.rand(B, 6, generator=gen, device=env.device) * 2 - 1 for _ in range(16)]
hist = {}; X=[];U=[];D=[];F=[]
for i in range(80):
    x_prev = env.x_T.clone(); ts = env.timestep_t.clone(); off = env.w_off_t.clone()
    env.step_tensor(acts[i % 16], want_obs=False)
    fl = env.step_flags_t
    idx = torch.nonzero((fl & 0xffff) != 0).flatten()
    for j in idx.tolist():
        f = int(fl[j]) & 0xffff; hist[(i, f)] = hist.get((i, f), 0) + 1
    if len(idx):
        X.append(x_prev[:, idx].t().double().cpu().numpy()); U.append(env.u_T[:, idx].t().double().cpu().numpy())
        D.append(env.weather_t[(off[idx] + ts[idx]).long()].double().cpu().numpy()); F.append(fl[idx].cpu().numpy())
print(hist)
print(env.metrics())
if X: np.savez("gpurun_out/flags_smallbatch.npz", X=np.concatenate(X), U=np.concatenate(U), D=np.concatenate(D), flags=np.concatenate(F))
