"""Throughput over time under sustained load (bench.py's step sequence, B = 65 536, fp32): chunks of 250 steps.  Shows the
boost-clock burst that a 20-step timed region measures and the level the GPU settles at.
    python tools/sustained_rate.py [scheme] [n_chunks] [n_sub|0] [cycled|fresh]
`cycled` (the default, as in rounds 3-5): 16 fixed action tensors in rotation -- every environment's Delta-u walk then has a constant drift per 16
steps and its controls end up pinned at 0 or 1 (a corner of the control cube): a STRESS workload.  `fresh`: new uniform actions every step, as bench.py."""
import sys, time
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np, torch
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
scheme = sys.argv[1] if len(sys.argv) > 1 else "rk4"
n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 40
n_sub = (int(sys.argv[3]) or None) if len(sys.argv) > 3 else None
fresh = len(sys.argv) > 4 and sys.argv[4] == "fresh"
B = 65536
w = synthetic_weather(35040); starts = np.arange(0, 35040 - 5760 - 60, 96)
env = TomatoVecEnv(B, weather=w, dtype="float32", scheme=scheme, n_sub=n_sub, season_length=60, start_rows=starts.tolist(),
                   start_days=(starts / 96.0).tolist(), seed=666)
env.reset_tensor()
g = torch.Generator(device=env.device).manual_seed(1)
acts = [torch.rand(B, 6, generator=g, device=env.device) * 2 - 1 for _ in range(16)]
for i in range(5): env.step_tensor(acts[i % 16])
torch.cuda.synchronize()
t_start = time.perf_counter()
for c in range(n_chunks):
    t0 = time.perf_counter()
    for i in range(250): env.step_tensor(torch.rand(B, 6, generator=g, device=env.device) * 2 - 1 if fresh else acts[i % 16])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    m = env.metrics()
    if c < 6 or c % 4 == 3:
        print(f"t = {time.perf_counter() - t_start:6.2f} s  chunk {c:3d}: {B * 250 / el:.3e} env-steps/s ({el / 250 * 1e3:.3f} ms/step); "
              f"guard retries {m['n_guard_retries']:.0f}, refined sub-steps {m['n_refined_substeps']:.0f}, failed {m['n_ode_fail']:.0f} of {m['n_env_steps']:.3e} env-steps", flush=True)
u = env.u[:B]
print("scheme", scheme, "n_sub", env.n_sub, "actions", "fresh" if fresh else "cycled", "ODE failures", env.metrics()["n_ode_fail"],
      "| controls at a bound (0 or 1) right now: %.1f %% of the batch's 6 x B values" % (100.0 * float(((u <= 0) | (u >= 1)).float().mean())))
