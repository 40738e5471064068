import os, sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/greenlight-gym2_amd')
from oracle import gl_oracle as O
G = lambda n: np.load(f'/root/repo/tests/golden/{n}.npz')
p = G('params_default')['p'].astype(np.float64)
R = G('rollout_10day'); w = R['weather']; X = R['X']; U = R['U']
from gl_gym_amd.greenlight_model import GreenLight
for layout in ('one', 'quad'):
    os.environ['GLGYM_LAYOUT'] = layout
    for dtype in ('float64', 'float32'):
        gl = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype) if 'dtype' in GreenLight.__init__.__code__.co_varnames else GreenLight(28,6,10,208,900.0)
        break
import torch
from gl_gym_amd.tomato_env import TomatoVecEnv
for dtype in ('float64', 'float32'):
    res = {}
    for layout in ('one', 'quad'):
        os.environ['GLGYM_LAYOUT'] = layout
        env = TomatoVecEnv(8, weather=w, dtype=dtype, n_sub=240, season_length=1, start_rows=[0], seed=5, auto_reset=False)
        env.reset()
        a = np.random.default_rng(0).uniform(-1, 1, (8, 6)).astype(np.float32)
        x0 = env.x.double().cpu().numpy().copy()
        env.step(a)
        res[layout] = env.x.double().cpu().numpy().copy()
        print(dtype, layout, 'flags', env.step_flags_t.cpu().numpy()[:4], 'done', env.done_t.cpu().numpy()[:4])
        u = env.u.double().cpu().numpy()
        env.close()
    ref = np.array([O.rk_sc_guarded(x0[b], u[b], w[0], p, 900.0, 240, 4, 2)[0] for b in range(8)])
    for layout in ('one', 'quad'):
        d = np.abs(res[layout] - ref) / np.maximum(np.abs(ref), 1e-6)
        print(dtype, layout, 'max rel diff vs oracle %.2e at state' % d.max(), np.unravel_index(d.argmax(), d.shape), 'per-state max', np.array2string(d.max(axis=0), precision=1))
