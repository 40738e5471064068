#!/usr/bin/env python3
"""Are two builds of libglgym.so bit-identical on the GPU?  (round 6: the policy refactor sc_policy.hpp must not change a bit.)

    GLGYM_LIB=/path/a.so python tools/lib_bitcompare.py dump /tmp/a.npz
    GLGYM_LIB=/path/b.so python tools/lib_bitcompare.py dump /tmp/b.npz
    python tools/lib_bitcompare.py compare /tmp/a.npz /tmp/b.npz

Workloads: the fused step in every kernel layout (fp32 one lane per environment, one- and two-wave builds; fp32 and fp64 four lanes per
environment), actions and verified raw controls, every scheme; glgym_evalF on raw-control-jump tuples (sequential and two-rungs ladder)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))


def dump(path):
    import torch
    from gl_gym_amd import GreenLight
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.utils import synthetic_weather
    w = synthetic_weather(n_rows=4000)
    w[:, 4] *= 2.5                                  # windy: refined windows, limiter, the odd flag
    out = {}
    starts = list(range(0, 2000, 37))
    for scheme in ("ls5", "rk4", "rk3", "rk2"):
        for tag, dtype, B, layout, occ in (("f32one", "float32", 20480, "one", 1), ("f32occ2", "float32", 20480, "one", 2),
                                           ("f32quad", "float32", 2048, "quad", 0), ("f64quad", "float64", 512, None, 0)):
            env = TomatoVecEnv(B, weather=w, dtype=dtype, scheme=scheme, season_length=1, start_rows=starts, seed=5, auto_reset=False)
            if layout:
                env.set_layout(layout)
            if occ:
                env.set_occupancy(occ)
            env.reset_tensor()
            g = torch.Generator(device=env.device).manual_seed(3)
            for k in range(4):
                env.step_tensor(torch.rand(B, 6, generator=g, device=env.device) * 2 - 1, want_obs=False)
            env.step_tensor(controls_t=torch.rand(B, 6, generator=g, device=env.device, dtype=env.tdtype), want_obs=False)
            out[f"{scheme}_{tag}_x"] = env.x.cpu().numpy()
            out[f"{scheme}_{tag}_flags"] = env.step_flags_t.cpu().numpy()
            out[f"{scheme}_{tag}_reward"] = env.reward_t[:B].cpu().numpy()
            env.close()
    g = np.load(ROOT / "tests" / "golden" / "step_tight_jump.npz")
    X, U, D = g["X"][:96], g["U"][:96], g["D"][:96]
    for dtype in ("float64", "float32"):
        for par in (True, False):
            m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset="throughput")
            m.set_ladder_parallel(par)
            try:
                out[f"evalF_{dtype}_{int(par)}"] = m.evalF_batch(X, U, D)
            except Exception as e:                   # a failed row raises: record which
                out[f"evalF_{dtype}_{int(par)}"] = np.array([hash(str(e)) % 1000])
            m.close()
    np.savez(path, **out)
    print("dumped", len(out), "arrays to", path)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    bad = 0
    for k in A.files:
        same = np.array_equal(A[k], B[k], equal_nan=True)
        if not same:
            bad += 1
            d = np.abs(A[k].astype(np.float64) - B[k].astype(np.float64))
            print(f"DIFFERENT {k}: {int((A[k] != B[k]).sum())} of {A[k].size} entries, max |d| {np.nanmax(d):.3e}")
    print(f"{len(A.files)} arrays compared, {bad} differ")
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2])
    else:
        sys.exit(1 if compare(sys.argv[2], sys.argv[3]) else 0)
