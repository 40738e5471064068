import sys
sys.path.insert(0, "greenlight-gym2_amd"); sys.path.insert(0, ".")
import numpy as np, torch
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd import GreenLight
g = np.load("tests/golden/rollout_10day.npz")
acts, w, XR = g["actions"], g["weather"], g["X"]
scheme = sys.argv[1] if len(sys.argv) > 1 else "rk2"
env = TomatoVecEnv(64, weather=w, dtype="float64", scheme=scheme, season_length=10, pred_horizon=0.5, auto_reset=False)
m = GreenLight(28, 6, 10, 208, 900.0, dtype="float64", scheme=scheme, n_sub=env.n_sub)
env.reset()
for k in range(len(acts)):
    xp = env.x[0].double().cpu().numpy().copy()
    a = torch.as_tensor(np.repeat(acts[k][None], 64, 0), device=env.device)
    env.step_tensor(a, want_obs=False)
    xn = env.x[0].double().cpu().numpy()
    u = env.u[0].double().cpu().numpy()
    ref = np.array(m.evalF(xp, u, w[k], env.p.astype(np.float64)))
    e = np.max(np.abs(xn - ref) / np.maximum(np.abs(ref), 1e-3))
    if e > 1e-9 or k % 100 == 0:
        print(k, "step vs evalF", e, "argmax", int(np.argmax(np.abs(xn - ref) / np.maximum(np.abs(ref), 1e-3))), env.metrics())
    if e > 1e-6:
        print("xp", xp.tolist()); print("u", u.tolist()); print("xn", xn.tolist()); print("ref", ref.tolist()); break
