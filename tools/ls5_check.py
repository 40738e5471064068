#!/usr/bin/env python3
"""Round-5 GPU check of the five-stage 2N scheme ("ls5") through the C ABI: evalF on the tight one-step / storm / jump fixtures against
tight truth and against the CPU checker's restatement (fp64 + fp32, throughput and parity presets), the 10-day fixture through
glgym_step, and a short timing of ls5 against rk4 at B = 65 536.      python tools/ls5_check.py [quick]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "greenlight-gym2_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from gl_gym_amd import GreenLight
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
from oracle import gl_oracle as O

G = lambda n: np.load(ROOT / "tests" / "golden" / f"{n}.npz")
p0 = G("params_default")["p"].astype(np.float64)
COLMAX = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5, 2.6e5, 6e4, 3.2e3, 60.])
sce = lambda a, b: np.abs(a - b) / np.maximum(np.abs(b), 1e-3 * COLMAX)

for name in ("step_tight", "step_tight_storm", "step_tight_jump"):
    g = G(name)
    X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
    P = g["P"].astype(np.float64) if "P" in g.files else None
    for preset, (n, w) in (("throughput", (128, 2)), ("parity", (192, 1))):
        ref = np.array([O.rk_sc_guarded(X[i], U[i], D[i], P[i] if P is not None else p0, 900.0, n, 5, w, verify=True)[0] for i in range(len(X))])
        for dtype in ("float64", "float32"):
            m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, scheme="ls5", preset=preset)
            t = time.time()
            try:
                got = m.evalF_batch(X, U, D, P if P is not None else p0)
                fail = 0
            except Exception as e:           # GLGYM_EODE
                got = np.full_like(X, np.nan); fail = str(e)
            dt_ = time.time() - t
            e_t = sce(got, XT).max(axis=1); e_o = sce(got, ref).max(axis=1)
            print(f"{name:18s} ls5 {preset:10s} {dtype}: vs tight max {np.nanmax(e_t):.2e} (n>1e-4: {(e_t > 1e-4).sum()}) | vs CPU checker max {np.nanmax(e_o):.2e} "
                  f"median {np.nanmedian(e_o):.1e} | failed {fail} | {dt_*1e3:.0f} ms", flush=True)
            m.close()

# 10-day rollout through glgym_step
g = G("rollout_10day")
acts, w, XR = g["actions"], g["weather"], g["X"]
for scheme, preset in (("ls5", "throughput"), ("ls5", "parity"), ("rk4", "throughput")):
    for dtype in ("float32", "float64"):
        for layout in (("one", "quad") if dtype == "float32" else ("auto",)):
            env = TomatoVecEnv(64, weather=w, dtype=dtype, scheme=scheme, preset=preset, season_length=(len(acts) - 1) // 96, pred_horizon=0.5, auto_reset=False)
            env.set_layout(layout)
            env.reset_tensor()
            a_all = torch.as_tensor(acts, device=env.device)
            Xs = [env.x[0].double().cpu().numpy()]
            for k in range(len(acts)):
                env.step_tensor(a_all[k][None].expand(64, 6).contiguous(), want_obs=False)
                Xs.append(env.x[0].double().cpu().numpy())
            mtr = env.metrics()
            print(f"10-day {scheme} {preset} n_sub {env.n_sub} window {env.window} {dtype} layout {layout}: {O.scaled_rel_err(np.array(Xs), XR):.2e} failed {mtr.get('n_ode_fail')} "
                  f"refined {mtr.get('n_refined_substeps')} retries {mtr.get('n_guard_retries')}", flush=True)
            env.close()

if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0)
# timing: bench workload, fresh actions
weather = synthetic_weather(n_rows=35040, dt=900.0, seed=2024)
starts = np.arange(0, 35040 - 5760 - 60, 96)
for B, dtype in ((65536, "float32"), (4096, "float64"), (16384, "float32"), (8, "float32"), (262144, "float32")):
    for scheme, n_sub, window in (("ls5", 128, 0), ("rk4", 240, 0), ("ls5", 192, 1), ("rk3", 270, 0)):
        env = TomatoVecEnv(B, weather=weather, dtype=dtype, scheme=scheme, n_sub=n_sub, window=window, season_length=60, pred_horizon=0.5, seed=666,
                           start_rows=starts, auto_reset=True)
        env.reset_tensor()
        env.x_T.mul_(1 + 1e-3 * torch.randn(env.x_T.shape, device=env.device, generator=torch.Generator(device=env.device).manual_seed(1234)).to(env.tdtype))
        gen = torch.Generator(device=env.device).manual_seed(666)
        def one():
            env.action_t.uniform_(-1.0, 1.0, generator=gen)
            env._launch_step(raw_control=False)
            env._launch_obs(env.obs_t)
            env._launch_reset(env.done_t)
            env._launch_obs(env.obs_t, env.done_t, env.term_obs_t)
        K = 300 if dtype == "float32" else 60
        for _ in range(30):
            one()
        env.metrics_t.zero_()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(K):
            one()
        torch.cuda.synchronize(); el = time.perf_counter() - t
        mtr = env.metrics()
        print(f"B {B} {dtype} {scheme} n_sub {n_sub} window {window or 'def'}: {B*K/el:.3e} env-steps/s, {1e3*el/K:.3f} ms/step; refined/env-step {mtr.get('n_refined_substeps',0)/(B*K):.3f} "
              f"retries {mtr.get('n_guard_retries')} failed {mtr.get('n_ode_fail')} flags err/branch/cap/heavy {mtr.get('n_flag_err')}/{mtr.get('n_flag_branch')}/{mtr.get('n_flag_cap')}/{mtr.get('n_flag_heavy')}", flush=True)
        env.close()
