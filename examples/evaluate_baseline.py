#!/usr/bin/env python3
"""Rule-based controller over one season, batched on the GPU -- the counterpart of the reference's
gl_gym/experiments/evaluate_baseline.py:12-37 (`evaluate_controller`) on this repo's API.

    python examples/evaluate_baseline.py --n-envs 1024 --season 10 [--uncertainty 0.2] [--weather-csv-dir DIR ...]

Per step the reference records obs[:23], reward and 8 info keys for ONE env; here the same 32 columns are recorded
for env 0 and, in addition, batch means over all envs (each env starts at a different day of the weather tensor).
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))

from gl_gym_amd.baseline import RuleBasedController          # noqa: E402
from gl_gym_amd.tomato_env import TomatoVecEnv               # noqa: E402
from gl_gym_amd.utils import load_weather_data, synthetic_weather   # noqa: E402
from gl_gym_amd import INFO_KEYS                              # noqa: E402

COLS = ["EPI", "revenue", "heat_cost", "co2_cost", "elec_cost", "temp_violation", "co2_violation", "rh_violation"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n-envs", type=int, default=1024)
    ap.add_argument("--season", type=float, default=10, help="season length [days]")
    ap.add_argument("--uncertainty", type=float, default=0.0, help="crop-parameter noise scale (stochastic mode)")
    ap.add_argument("--dtype", default="float32")
    ap.add_argument("--weather-csv-dir", default=None, help="reference-style weather dir (else synthetic weather)")
    ap.add_argument("--location", default="Bleiswijk")
    ap.add_argument("--source", default="GL")
    ap.add_argument("--year", type=int, default=2009)
    args = ap.parse_args()

    if args.weather_csv_dir:
        w = load_weather_data(args.weather_csv_dir, args.location, args.source, args.year, 0, args.season, 49, 900, 10)
        starts, days = [0], [0.0]
    else:
        w = synthetic_weather(n_rows=35040)
        starts = list(range(0, 35040 - int(args.season * 96) - 60, 96))
        days = [s / 96.0 for s in starts]
    env = TomatoVecEnv(args.n_envs, weather=w, dtype=args.dtype, season_length=args.season, start_rows=starts,
                       start_days=days, uncertainty_scale=args.uncertainty, seed=666, auto_reset=False)
    ctrl = RuleBasedController()
    N1 = env.N + 1
    rec0 = np.zeros((N1, 23 + 1 + len(COLS)))
    mean = np.zeros((N1, 1 + len(COLS)))
    idx = [INFO_KEYS.index(k) for k in COLS]
    obs = env.reset_tensor()
    t0 = time.time()
    for k in range(N1):
        obs, rew, done, info = env.step_tensor(controller=ctrl)     # glgym_rule_based + glgym_step, no host round trip
        rec0[k, :23] = obs[0, :23].cpu().numpy()
        rec0[k, 23] = float(rew[0])
        rec0[k, 24:] = info[idx, 0].cpu().numpy()
        mean[k, 0] = float(rew.mean())
        mean[k, 1:] = info[idx].mean(dim=1).cpu().numpy()
    el = time.time() - t0
    assert bool(done.all())
    print(f"{args.n_envs} envs x {N1} steps in {el:.2f} s ({args.n_envs * N1 / el:.3e} env-steps/s incl. controller + host loop)")
    print("batch-mean cumulative reward %.4f, EPI %.4f EUR/m2, ODE failures %d" %
          (mean[:, 0].sum(), mean[:, 1].sum(), env.metrics()["n_ode_fail"]))
    print("env 0 cumulative:", {c: round(float(rec0[:, 24 + i].sum()), 5) for i, c in enumerate(COLS)})
    env.close()


if __name__ == "__main__":
    main()
