#!/usr/bin/env python3
"""Storm study (round 2): where does the fixed-step sub-stepper return finite-but-wrong states, and which a-priori
quantity predicts it?  TEST INFRASTRUCTURE (uses the oracle).

    python oracle/studies/storm_proto.py [n_tuples]

For random (wind 15-35 m/s, tOut -5..15 C, vents 0.7-1, other controls random) tuples: spin the reset state up for 1800 s
with a tight solve, take one 900 s step with (a) Radau 1e-11 (truth), (b) the oracle restatement of the product scheme
at n_sub 320 / 640; print the scaled error, the spectral radius of the Jacobian along the tight trajectory, and the
candidate rate bounds.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
from scipy.integrate import solve_ivp

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))

from oracle import gl_oracle as O  # noqa: E402
from gl_gym_amd.parameters import init_default_params  # noqa: E402
from gl_gym_amd.utils import synthetic_weather, init_state  # noqa: E402


def jac(x, u, d, p):
    f0 = O.rhs(x, u, d, p)
    J = np.empty((28, 28))
    for j in range(28):
        h = 1e-6 * max(abs(x[j]), 1e-2)
        xp = x.copy(); xp[j] += h
        xm = x.copy(); xm[j] -= h
        J[:, j] = (O.rhs(xp, u, d, p) - O.rhs(xm, u, d, p)) / (2 * h)
    return J, f0


def tight(x, u, d, p, dt, dense=False):
    s = solve_ivp(lambda t, y: O.rhs(y, u, d, p), (0.0, dt), x, method="Radau", rtol=1e-11, atol=1e-11,
                  dense_output=dense)
    assert s.success
    return s


def scaled(a, b):
    sc = np.maximum(np.abs(b), 1e-3 * np.abs(b))
    sc[sc == 0] = 1.0
    return np.abs(a - b) / np.maximum(np.abs(b), 1e-3)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    p = init_default_params().astype(np.float64)
    w = synthetic_weather(n_rows=4000)
    rng = np.random.default_rng(7)
    bad = 0
    for i in range(n):
        d = w[int(rng.integers(0, 3000))].copy()
        d[4] = rng.uniform(15, 35)
        d[1] = rng.uniform(-5, 15)
        d[5] = d[1] - rng.uniform(5, 20)
        u = rng.uniform(0, 1, 6)
        u[3] = rng.uniform(0.7, 1.0)
        if rng.uniform() < 0.5:
            u[5] = 0.0
        x0 = init_state(d)
        xs = tight(x0, u, d, p, 1800.0).y[:, -1]
        s = tight(xs, u, d, p, 900.0, dense=True)
        xt = s.y[:, -1]
        lam = 0.0
        for t in (0.0, 5.0, 30.0, 120.0, 450.0, 900.0):
            J, _ = jac(s.sol(t), u, d, p)
            ev = np.linalg.eigvals(J)
            lam = max(lam, float(np.max(-ev.real)))
        e320 = scaled(O.rk_lagged(xs, u, d, p, 900.0, 320, 4, 2), xt)
        e640 = scaled(O.rk_lagged(xs, u, d, p, 900.0, 640, 4, 2), xt)
        flag = "BAD" if e320.max() > 1e-4 else "ok "
        bad += e320.max() > 1e-4
        print(f"{i:3d} wind {d[4]:5.1f} tOut {d[1]:5.1f} u {np.round(u, 2)} lam_max {lam:6.3f} h*lam {lam * 900 / 320:5.2f} "
              f"e320 {e320.max():8.1e} (x{int(e320.argmax())}) e640 {e640.max():8.1e} {flag}", flush=True)
    print("bad:", bad, "of", n)


if __name__ == "__main__":
    main()
