#!/usr/bin/env python3
"""Round 5: the window length follows the rate bound (gl_oracle.c rk_sc_impl, gl_sc_varwin) -- go / no-go on the CPU before the kernels.

r05_heavy_tuples.npz: the 396 env-steps of 2.6e7 on which the round's first ls5 kernel took >= 40 sub-steps beyond the nominal 128
(tools/flag_tuples.py with GLGYM_TOOL_HEAVY=40 on the bench workload: one per launch -- at one wave per SIMD the launch waits for them).
Two kinds: (a) lanes whose rate bound sits 5-8 % over the nominal limit for the whole env-step while the smooth bound at x0 (all the
pre-pass looked at) was below it: 3 sub-steps per window instead of 2, +50 % for a 6 % excess; (b) bursts: a wet surface pinned at 77 1/s
in the FIRST window only -- 128 sub-steps of 0.11 s through a 14 s window whose transient is over after the first second.

Per tuple: sub-steps and windows under round 4's rule (gl_sc_varwin = 0) and the new one, the error of both against RK4-4096, and
the GPU time the counts stand for (3.37 us per five-stage sub-step, 2.0 us per window: tools/window_cost.py).
    PYTHONPATH=.:greenlight-gym2_amd python oracle/studies/varwin_study.py > oracle/studies/varwin_study_result.txt"""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "greenlight-gym2_amd")); sys.path.insert(0, str(ROOT / "tests"))
from oracle import gl_oracle as O
from gl_gym_amd.parameters import init_default_params
from test_jump_fixture import sce, judge

p = init_default_params().astype(np.float64)
lib = O.lib(); lib.gl_oracle_last_windows.restype = C.c_double
varwin = C.c_int.in_dll(lib, "gl_sc_varwin")
d = np.load(ROOT / "oracle" / "studies" / "r05_heavy_tuples.npz")
X, U, D, F = d["X"], d["U"], d["D"], d["flags"]
extra = F >> 16
scale = np.maximum(np.abs(X).max(axis=0), 1e-3)
rows = []
for i in range(len(X)):
    varwin.value = 1
    truth = O.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, 4096, 4, 4)[0]
    out = [int(extra[i])]
    for f in (0, 1):
        varwin.value = f
        r, ret, ref, failed = O.rk_sc_guarded(X[i], U[i], D[i], p, 900.0, 128, 5, 2)
        out += [128 + ref, lib.gl_oracle_last_windows(), float(np.max(np.abs(r - truth) / np.maximum(np.abs(truth), scale))), ret]
    rows.append(out)
varwin.value = 1
rows = np.array(rows)
base = 128 * 3.37 + 64 * 2.0
t_old, t_new = rows[:, 1] * 3.37 + rows[:, 2] * 2.0 - base, rows[:, 5] * 3.37 + rows[:, 6] * 2.0 - base
print(f"{len(X)} heavy env-steps of the bench workload (>= 40 extra sub-steps on the GPU under round 4's rule)")
print("extra time per env-step [us], old rule -> window length follows the bound: mean %.0f -> %.0f, median %.0f -> %.0f, max %.0f -> %.0f"
      % (t_old.mean(), t_new.mean(), np.median(t_old), np.median(t_new), t_old.max(), t_new.max()))
print("max scaled error against RK4-4096: old %.2e, new %.2e; extra attempts old %d, new %d" % (rows[:, 3].max(), rows[:, 7].max(), rows[:, 4].sum(), rows[:, 8].sum()))
for lo, hi in ((40, 64), (64, 72), (72, 126), (126, 127), (127, 10000)):
    m = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
    if m.any():
        print("   GPU extra %4d-%-5d: %3d tuples, sub-steps %5.0f -> %5.0f, windows %5.1f -> %5.1f, extra us %4.0f -> %4.0f"
              % (lo, hi - 1, m.sum(), rows[m, 1].mean(), rows[m, 5].mean(), rows[m, 2].mean(), rows[m, 6].mean(), t_old[m].mean(), t_new[m].mean()))
# the raw-jump fixture, UNVERIFIED (beyond the action path's spec; the margin the old rule's accidentally short storm windows gave)
g = np.load(ROOT / "tests" / "golden" / "step_tight_jump.npz")
for order, win, n in ((5, 2, 128), (4, 4, 240)):
    for f in (0, 1):
        varwin.value = f
        res = [O.rk_sc_guarded(g["X"][i], g["U"][i], g["D"][i], p, 900.0, n, order, win) for i in range(len(g["X"]))]
        out = np.array([r[0] for r in res])
        e = sce(out, g["X_tight"]).max(axis=1)
        wv = judge(np.array([O.rk_sc_guarded(g["X"][i], g["U"][i], g["D"][i], p, 900.0, n, order, win, verify=True)[0] for i in range(len(g["X"]))]), g["X_tight"], 1e-4)
        print("jump fixture, order %d n_sub %d, %s: unverified > 1e-4 %d (max %.2e, q99 %.1e), mean sub-steps %.0f, extra attempts %d; verified: wrong %d, floor %d"
              % (order, n, "window follows the bound" if f else "round 4's rule", int((e > 1e-4).sum()), e.max(), np.quantile(e, 0.99),
                 n + np.mean([r[2] for r in res]), sum(r[1] for r in res), wv[0], wv[1]))
varwin.value = 1
