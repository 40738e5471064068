#!/usr/bin/env python3
"""One-at-a-time sensitivity of the sub-stepper's tunables (VERDICT r05 "next 3"): is any of them sitting on a cliff?

For every `#define SC_* value` of greenlight-gym2_amd/csrc/sc_policy.hpp, x 0.8 and x 1.2 (integers: the neighbouring values), the HOST
instantiation of the product's gl_model.hpp (tests/hostmath/hostmath.cpp, fp64) is rebuilt from a COPY of the headers with that one
value changed, and run over the tuning fixtures AND the round-6 hold-outs with the shipped scheme (ls5, throughput preset 128 / window
2; verified where the kernels verify).  Recorded per variant and fixture: max scaled error vs the tight truth, tuples / steps above 1e-4
that are not at the 0 C metric floor, failed integrations, and the mean number of sub-steps per env-step (the cost side).  Nothing is
asserted: the table is the evidence.   CPU only, ~5 minutes on 8 cores.

    python tools/tunable_sensitivity.py > profiles/r06_tunable_sensitivity.txt
"""
import ctypes
import re
import shutil
import subprocess
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "greenlight-gym2_amd" / "csrc"
sys.path.insert(0, str(ROOT / "tests"))
sys.path.insert(0, str(ROOT / "greenlight-gym2_amd"))

SKIP = {"SC_ATTEMPTS": "the ladder's length is structural (n, 2n, 4n, 8n; static_assert in the two-rungs-at-a-time ladder)",
        "SC_GROW": "2.0001 = 'at most doubling' with a rounding guard, not a magnitude"}
INT = {"SC_MAX_REFINE": (32, 128), "SC_HEAVY": (2, 4)}


def tunables():
    txt = (CSRC / "sc_policy.hpp").read_text()
    return re.findall(r"^#define (SC_[A-Z_]+) (\S+)$", txt, flags=re.M)


def build(name, value):
    d = Path(tempfile.mkdtemp(prefix="glsens_"))
    for f in ("gl_model.hpp", "sc_policy.hpp"):
        shutil.copy(CSRC / f, d / f)
    if name is not None:
        p = d / "sc_policy.hpp"
        txt, n = re.subn(rf"^#define {name} \S+$", f"#define {name} {value}", p.read_text(), flags=re.M)
        assert n == 1
        p.write_text(txt)
    so = d / "libhostmath.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", f"-I{d}", "-o", str(so),
                           str(ROOT / "tests" / "hostmath" / "hostmath.cpp")])
    return so


def evaluate(args):
    name, value = args
    from conftest import judge_rollout
    so = build(name, value)
    lib = ctypes.CDLL(str(so))
    dp = ctypes.POINTER(ctypes.c_double)
    P = lambda a: a.ctypes.data_as(dp)  # noqa: E731

    def step(x, u, d, p, dt, n_sub, win, verify):
        out, st = np.empty(28), np.zeros(2)
        x, u, d, p = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, u, d, p)]
        r = lib.hostmath_step_guarded2(P(x), P(u), P(d), P(p), 0, ctypes.c_double(dt), n_sub, 5, win, int(verify), P(out), P(st))
        return out, r, int(st[0]), bool(st[1])

    G = ROOT / "tests" / "golden"
    p0 = np.load(G / "params_default.npz")["p"].astype(np.float64)
    colmax = np.array([1500, 1500, 30, 30, 30, 30, 30, 30, 30, 60, 30, 30, 30, 30, 30, 3000, 3000, 60, 30, 30, 30, 30, 2e4, 1e5, 2.6e5, 6e4, 3.2e3, 60.])
    res = {}
    # one-step fixtures (tuples): scaled error of tests/test_jump_fixture.py, floor rule 1e-4 K
    for fx, verify in (("step_tight", False), ("step_tight_storm", False), ("step_tight_jump", True)):
        g = np.load(G / f"{fx}.npz")
        X, U, D, XT = g["X"], g["U"], g["D"], g["X_tight"]
        worst, above, failed, extra = 0.0, 0, 0, 0
        for i in range(len(X)):
            p = g["P"][i].astype(np.float64) if "P" in g.files else p0
            y, r, ex, bad = step(X[i], U[i], D[i], p, 900.0, 128, 2, verify)
            failed += bad
            extra += ex
            if bad:
                continue
            e = np.abs(y - XT[i]) / np.maximum(np.abs(XT[i]), 1e-3 * colmax)
            fl = (e > 1e-4) & (np.abs(y - XT[i]) < 1e-4) & (np.arange(28) < 22) & (np.abs(XT[i]) < 1.0)
            worst = max(worst, float(e.max()))
            above += int(((e > 1e-4) & ~fl).any())
        res[fx] = (worst, above, failed, 128 * (3 if verify else 1) + extra / len(X))
    # rollouts (free-running)
    for fx, dt, verify, stride in (("rollout_10day", 900.0, False, 1), ("holdout_gl2010_random", 900.0, False, 1),
                                   ("holdout_gl2010_rulebased", 900.0, True, 1), ("holdout_runtime_dt300", 300.0, True, 3)):
        g = np.load(G / f"{fx}.npz")
        w, XR, U = g["weather"], g["X"], g["U"].astype(np.float64)
        p = (g["p"] if "p" in g.files else p0).astype(np.float64)
        x = (g["x0"] if "x0" in g.files else XR[0]).copy()
        n_sub = max(2, int(-(-(128 * dt / 900.0) // 2) * 2))
        Xs, failed, extra = [x.copy()], 0, 0
        for k in range(len(U)):
            x, r, ex, bad = step(x, U[k], w[k], p, dt, n_sub, 2, verify)
            failed += bad
            extra += ex
            if (k + 1) % stride == 0:
                Xs.append(x.copy())
        plain, who, row, real, floor = judge_rollout(np.array(Xs), XR[:len(Xs)], abs_floor=1e-4)
        res[fx] = (plain, real, failed, n_sub * (3 if verify else 1) + extra / len(U))
    shutil.rmtree(so.parent, ignore_errors=True)
    return name, value, res


def main():
    tn = tunables()
    jobs = [(None, None)]
    for name, val in tn:
        if name in SKIP:
            continue
        if name in INT:
            jobs += [(name, str(v)) for v in INT[name]]
        else:
            v = float(val)
            jobs += [(name, repr(0.8 * v)), (name, repr(1.2 * v))]
            if name == "SC_SAFETY":      # x 1.2 = 1.104 steps BEYOND the stability interval (unstable by construction): also the interval's edge itself
                jobs += [(name, "0.99")]
    with ProcessPoolExecutor(8) as ex:
        out = list(ex.map(evaluate, jobs))
    fixtures = list(out[0][2])
    print("# One-at-a-time sensitivity of the sub-stepper's tunables (tools/tunable_sensitivity.py; host fp64 instantiation of the shipped")
    print("# headers, scheme ls5 at the throughput preset 128 / window 2, verified where the kernels verify).  Per fixture:")
    print("#   max scaled error | tuples (one-step fixtures) or steps (rollouts) above 1e-4 away from the 0 C floor | failed | mean sub-steps per env-step")
    print("# Shipped values first; then each tunable x 0.8 and x 1.2 (integers: neighbours), everything else as shipped.")
    print("# SC_SAFETY is a fraction of the scheme's stability interval: 1.104 (x 1.2) lets sub-steps exceed it -- unstable by construction, listed")
    print("# for completeness; 0.99 (the interval's edge) is the meaningful upper neighbour.")
    for name, why in SKIP.items():
        print(f"# not perturbed: {name} -- {why}")
    shipped = dict(tn)
    print("%-16s %-10s " % ("tunable", "value") + " ".join("%-34s" % f for f in fixtures))
    for name, value, res in out:
        label = ("(as shipped)", "") if name is None else (name, "%s -> %.6g" % (shipped[name], float(value)))
        print("%-16s %-22s " % label + " ".join("%.2e %3d %2d %7.1f%14s" % (*res[f], "") for f in fixtures))
    base = out[0][2]
    worst = max((res[f][0] / max(base[f][0], 1e-12), name, value, f) for name, value, res in out[1:] for f in fixtures)
    newfail = [(name, value, f) for name, value, res in out[1:] for f in fixtures if res[f][2] > base[f][2] or res[f][1] > base[f][1]]
    print("# largest growth of a fixture's max error under any single perturbation: x%.2f (%s = %s on %s)" % worst)
    print("# perturbations that ADD a failed integration or a tuple / step above the bar: %s" % (newfail if newfail else "none"))


if __name__ == "__main__":
    main()
