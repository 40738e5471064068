"""A/B of the verified fp64 glgym_evalF call (the GreenLight() default: parity preset, two rungs at a time) between builds:
    GLGYM_LIB=tools/_libX.so python tools/evalf_ab_fp64.py        (one line per configuration; same tuples in every process)"""
import os, sys, time
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state
w = synthetic_weather(2000)
for dtype, preset in (("float64", "parity"), ("float64", "throughput"), ("float32", "parity")):
    for par, verify in ((True, "auto"), (False, "auto"), (True, "never")):
        rng = np.random.default_rng(11)
        m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset=preset); m.set_verify(verify); m.set_ladder_parallel(par)
        for B in (1, 8):
            D = w[rng.integers(0, len(w), B)]; X = np.array([init_state(d) for d in D]); U = rng.uniform(0, 1, (B, 6))
            for _ in range(20): Y = m.evalF_batch(X, U, D)
            t = []
            for _ in range(100):
                t0 = time.perf_counter(); m.evalF_batch(X, U, D); t.append((time.perf_counter() - t0) * 1e6)
            print(os.environ.get("GLGYM_LIB", "default").split("/")[-1], dtype, preset, "pair" if par else "seq ", verify, B, "%.1f us" % np.median(t),
                  "checksum %.17g" % float(np.sum(Y)), flush=True)
        m.close()
