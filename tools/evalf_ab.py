import sys, time, os
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state
w = synthetic_weather(2000); rng = np.random.default_rng(7)
def tuples(B):
    D = w[rng.integers(0, len(w), B)]; X = np.array([init_state(d) for d in D]); U = rng.uniform(0, 1, (B, 6)); return X, U, D
for dtype, preset in (("float32", "throughput"), ("float64", "parity")):
    for par in (True, False):
        for verify in ("auto", "never"):
            m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset=preset); m.set_verify(verify); m.set_ladder_parallel(par)
            for B in (1, 8):
                X, U, D = tuples(B)
                for _ in range(20): m.evalF_batch(X, U, D)
                t = []
                for _ in range(150):
                    t0 = time.perf_counter(); m.evalF_batch(X, U, D); t.append((time.perf_counter() - t0) * 1e6)
                print(os.environ.get("GLGYM_LIB", "default").split("/")[-1], dtype, preset, "pair" if par else "seq ", verify, B, "%.1f us" % np.median(t), flush=True)
            m.close()
