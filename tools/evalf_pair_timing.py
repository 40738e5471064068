"""Verified (two-rungs-at-a-time) glgym_evalF call, B = 1: median of 100 calls for fp64 parity / fp64 throughput / fp32 parity / fp32 throughput.
    GLGYM_LIB=... python tools/evalf_pair_timing.py"""
import os, sys, time
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state
w = synthetic_weather(2000)
res = []
for dtype, preset in (("float64", "parity"), ("float64", "throughput"), ("float32", "parity"), ("float32", "throughput")):
    rng = np.random.default_rng(11)
    m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset=preset); m.set_verify("auto"); m.set_ladder_parallel(True)
    D = w[rng.integers(0, len(w), 1)]; X = np.array([init_state(d) for d in D]); U = rng.uniform(0, 1, (1, 6))
    for _ in range(20): m.evalF_batch(X, U, D)
    t = []
    for _ in range(100):
        t0 = time.perf_counter(); m.evalF_batch(X, U, D); t.append((time.perf_counter() - t0) * 1e6)
    res.append("%s/%s %.0f" % (dtype[-2:], preset[:3], np.median(t)))
    m.close()
print(os.environ.get("GLGYM_LIB", "default").split("/")[-1], " | ".join(res), flush=True)
