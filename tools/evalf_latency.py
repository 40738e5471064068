#!/usr/bin/env python3
"""Latency of one glgym_evalF call through the C ABI -- the FFI seam the reference's GreenLight.evalF crosses once per
env.step() (/root/reference gl_gym/environments/tomato_env.py:142-163 -> models/greenlight_model.py evalF).
Host buffers in, host buffers out: the call includes H2D of (x, u, d[, p]), the kernel and D2H of x(dt), i.e. what a binding of
the reference pays per environment step.  Median / p10 / p90 of `reps` calls after a warm-up, per batch size, dtype, preset and
verification mode.        python tools/evalf_latency.py [reps] > profiles/r05_evalf_latency.txt"""
import sys, time
import numpy as np
sys.path.insert(0, "greenlight-gym2_amd")
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
w = synthetic_weather(2000)
rng = np.random.default_rng(7)


def tuples(B):
    D = w[rng.integers(0, len(w), B)]
    X = np.array([init_state(d) for d in D])
    U = rng.uniform(0, 1, (B, 6))
    return X, U, D


print("# glgym_evalF latency through the C ABI (host pointers in / out), MI355X; microseconds per CALL: median [p10, p90]")
print("# dtype  preset      scheme n_sub window verify   B    us/call            us/env-step")
for dtype in ("float64", "float32"):
    for preset in ("parity", "throughput"):
        for verify in ("auto", "never"):
            m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset=preset)
            m.set_verify(verify)
            for B in (1, 8, 64, 1024):
                X, U, D = tuples(B)
                for _ in range(20):
                    m.evalF_batch(X, U, D)
                t = []
                for _ in range(reps):
                    t0 = time.perf_counter()
                    m.evalF_batch(X, U, D)
                    t.append((time.perf_counter() - t0) * 1e6)
                t = np.array(t)
                print(f"{dtype:8s} {preset:10s} {m.scheme:5s} {m.n_sub:5d} {m.window:5d}  {verify:6s} {B:5d}   "
                      f"{np.median(t):8.1f} [{np.quantile(t, .1):7.1f}, {np.quantile(t, .9):7.1f}]   {np.median(t) / B:9.2f}")
            m.close()
# the evalF the reference calls (one tuple, parameters passed with the call): p given -> one more 1.7 kB H2D copy
from gl_gym_amd.parameters import init_default_params
p = init_default_params(208)
for dtype in ("float64", "float32"):
    m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype)
    X, U, D = tuples(1)
    for _ in range(20):
        m.evalF(X[0], U[0], D[0], p)
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); m.evalF(X[0], U[0], D[0], p); t.append((time.perf_counter() - t0) * 1e6)
    print(f"# GreenLight.evalF(x, u, d, p) as the reference calls it ({dtype}, defaults {m.scheme}/{m.n_sub}/window {m.window}): "
          f"median {np.median(t):.1f} us [{np.quantile(t, .1):.1f}, {np.quantile(t, .9):.1f}]")
    m.close()
