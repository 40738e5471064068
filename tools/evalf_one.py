#!/usr/bin/env python3
"""30 verified glgym_evalF calls (B = 1) under the profiler: python tools/evalf_one.py float64|float32 parity|throughput [seq]"""
import sys
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np
from gl_gym_amd import GreenLight
from gl_gym_amd.utils import synthetic_weather, init_state
dtype, preset = sys.argv[1], sys.argv[2]
w = synthetic_weather(2000); rng = np.random.default_rng(11)
m = GreenLight(28, 6, 10, 208, 900.0, dtype=dtype, preset=preset); m.set_verify("auto"); m.set_ladder_parallel(not (len(sys.argv) > 3 and sys.argv[3] == "seq"))
D = w[rng.integers(0, len(w), 1)]; X = np.array([init_state(d) for d in D]); U = rng.uniform(0, 1, (1, 6))
for _ in range(30): m.evalF_batch(X, U, D)
m.close()
