"""ctypes wrapper around oracle/_build/libgl_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product package (greenlight-gym2_amd/) never does.

Restates (CPU, fp64): aux_states.hpp:5-1271, ode.hpp:6-124, greenlight_model.cpp:46-120.
Parity status: RHS pinned by tests/golden/rhs_kat.npz; CVODES integrator "parity unpinned"
(bounded against a tight stiff solve instead) -- see oracle/gl_oracle.c header.
"""
from __future__ import annotations

import ctypes
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = _HERE / "_build" / "libgl_oracle.so"

NX, NU, ND, NP, NAUX = 28, 6, 10, 208, 239

_dp = ctypes.POINTER(ctypes.c_double)


def build(force: bool = False) -> Path:
    if force or not _SO.exists() or _SO.stat().st_mtime < (_HERE / "gl_oracle.c").stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(str(_SO))
        L.gl_oracle_aux.argtypes = [_dp] * 5
        L.gl_oracle_rhs.argtypes = [_dp] * 6
        L.gl_oracle_rk4.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rk4_split.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rhs_pipe.argtypes = [_dp] * 6
        L.gl_oracle_rk4_split_pipe.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rk4_lagged.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rk4_lagged_pipe.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rk_lagged.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp]
        L.gl_oracle_rk_lagged_pipe.argtypes = L.gl_oracle_rk_lagged.argtypes
        L.gl_oracle_rk_sc.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, _dp, _dp]
        L.gl_rate_bound.argtypes = [_dp] * 4
        L.gl_oracle_rk_sc_guarded.argtypes = [_dp] * 4 + [ctypes.c_double] + [ctypes.c_int] * 4 + [_dp, _dp]
        L.gl_oracle_rk_sc_guarded.restype = ctypes.c_int
        L.gl_oracle_rk_sc_guarded2.argtypes = [_dp] * 4 + [ctypes.c_double] + [ctypes.c_int] * 5 + [_dp, _dp]
        L.gl_oracle_rk_sc_guarded2.restype = ctypes.c_int
        L.gl_rate_bound.restype = ctypes.c_double
        L.gl_oracle_rk4_guarded.argtypes = [_dp] * 4 + [ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_rk4_guarded.restype = ctypes.c_int
        L.gl_oracle_rk4_batch.argtypes = [_dp] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int, _dp]
        L.gl_oracle_stiff.argtypes = [_dp] * 4 + [ctypes.c_double] * 3 + [_dp, ctypes.POINTER(ctypes.c_long)]
        L.gl_oracle_stiff.restype = ctypes.c_long
        L.gl_oracle_bdf.argtypes = [_dp] * 4 + [ctypes.c_double] * 3 + [_dp, _dp]
        L.gl_oracle_bdf.restype = ctypes.c_long
        L.gl_oracle_bdf_batch.argtypes = [_dp] * 4 + [ctypes.c_int] + [ctypes.c_double] * 3 + [_dp]
        L.gl_oracle_bdf_batch.restype = ctypes.c_long
        _lib = L
    return _lib


def _c(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if n is not None:
        assert a.size == n, (a.shape, n)
    return a


def _p(a):
    return a.ctypes.data_as(_dp)


def rhs(x, u, d, p, want_aux=False):
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    dx = np.empty(NX)
    aux = np.empty(NAUX)
    lib().gl_oracle_rhs(_p(x), _p(u), _p(d), _p(p), _p(dx), _p(aux))
    return (dx, aux) if want_aux else dx


def rk4(x, u, d, p, dt=900.0, n_sub=256):
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    lib().gl_oracle_rk4(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), _p(out))
    return out


def rk4_split(x, u, d, p, dt=900.0, n_sub=256):
    """Strang-split exact harvest flow + RK4 of the remaining RHS, every auxiliary evaluated at every stage
    (the kernels' scheme without its slow-auxiliary lag; see gl_oracle.c)."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    lib().gl_oracle_rk4_split(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), _p(out))
    return out


def rhs_pipe(x, u, d14, p):
    """ODE_pipe (ode.hpp:126-263); d14 = the 10 disturbances + tPipe, tGroPipe, pipeSwitchOff, groPipeSwitchOff."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d14, 14), _c(p, NP)
    dx = np.empty(NX)
    aux = np.empty(NAUX)
    lib().gl_oracle_rhs_pipe(_p(x), _p(u), _p(d), _p(p), _p(dx), _p(aux))
    return dx


def rk4_split_pipe(x, u, d14, p, dt=300.0, n_sub=256):
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d14, 14), _c(p, NP)
    out = np.empty(NX)
    lib().gl_oracle_rk4_split_pipe(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), _p(out))
    return out


def rk4_lagged(x, u, d, p, dt=900.0, n_sub=256, pipe=False):
    """Strang-split RK4 with the slow auxiliaries (tCan24, cLeaf, tCanSum) lagged to the sub-step start (gl_oracle.c)."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, 14 if pipe else ND), _c(p, NP)
    out = np.empty(NX)
    (lib().gl_oracle_rk4_lagged_pipe if pipe else lib().gl_oracle_rk4_lagged)(_p(x), _p(u), _p(d), _p(p), float(dt),
                                                                               int(n_sub), _p(out))
    return out


def rk_lagged(x, u, d, p, dt=900.0, n_sub=256, order=4, window=1, pipe=False):
    """The kernels' fixed-step schemes without the stability control: RK order 2 / 3 / 4 with tier 2b and the harvest flow
    shared by `window` sub-steps (the kernels: RK4 window 4, three-stage scheme window 3, midpoint window 4)."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, 14 if pipe else ND), _c(p, NP)
    out = np.empty(NX)
    (lib().gl_oracle_rk_lagged_pipe if pipe else lib().gl_oracle_rk_lagged)(
        _p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), int(order), int(window), _p(out))
    return out


def rk_sc(x, u, d, p, dt=900.0, n_sub=240, order=4, window=4):
    """The kernels' stability-controlled sub-stepper (gl_oracle.c rk_sc_impl).  Returns (x_next, stats) with
    stats = [sub-steps taken, max error-estimate ratio, max rate bound, flags]."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    st = np.zeros(4)
    lib().gl_oracle_rk_sc(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), int(order), int(window), _p(out), _p(st))
    return out, st


def rk_sc_guarded(x, u, d, p, dt=900.0, n_sub=240, order=4, window=4, pipe=False, verify=False, want_flags=False):
    """rk_sc with the kernels' guard (step-doubling ladder n, 2n, 4n, 8n).  Returns (x_next, retries, refined sub-steps beyond
    n_sub, failed) and, with want_flags, the kernels' step_flags word (include/glgym.h GLGYM_SF_*) as a fifth entry."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, 14 if pipe else ND), _c(p, NP)
    out = np.empty(NX)
    st = np.zeros(3)
    r = lib().gl_oracle_rk_sc_guarded2(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), int(order), int(window),
                                       int(bool(pipe)), int(bool(verify)), _p(out), _p(st))
    return (out, int(r), int(st[1]), bool(st[0])) + ((int(st[2]),) if want_flags else ())


def rate_bound(x, u, d, p):
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    return float(lib().gl_rate_bound(_p(x), _p(u), _p(d), _p(p)))


def rk4_guarded(x, u, d, p, dt=900.0, n_sub=256):
    """ROUND-1 scheme: fixed-step rk4_lagged with the non-finite-only guard (retry with 2x / 4x sub-steps); kept for
    regression comparisons -- the kernels run rk_sc_guarded.  Returns (x_next, retries)."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    r = lib().gl_oracle_rk4_guarded(_p(x), _p(u), _p(d), _p(p), float(dt), int(n_sub), _p(out))
    return out, int(r)


def rk4_batch(X, U, D, P, dt=900.0, n_sub=256):
    X, U, D = _c(X), _c(U), _c(D)
    B = X.shape[0]
    assert X.shape == (B, NX) and U.shape == (B, NU) and D.shape == (B, ND)
    P = _c(P)
    per_env = int(P.ndim == 2)
    assert P.shape == ((B, NP) if per_env else (NP,))
    out = np.empty((B, NX))
    lib().gl_oracle_rk4_batch(_p(X), _p(U), _p(D), _p(P), per_env, B, float(dt), int(n_sub), _p(out))
    return out


def stiff(x, u, d, p, dt=900.0, rtol=1e-6, atol=1e-6):
    """Adaptive implicit step map (extrapolated linearly-implicit Euler). Returns (x_next, n_rhs_evals)."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    ns = ctypes.c_long(0)
    nfev = lib().gl_oracle_stiff(_p(x), _p(u), _p(d), _p(p), float(dt), float(rtol), float(atol), _p(out),
                                 ctypes.byref(ns))
    if nfev < 0:
        raise RuntimeError("gl_oracle_stiff: step size underflow")
    return out, int(nfev)


def bdf(x, u, d, p, dt=900.0, rtol=1e-6, atol=1e-6):
    """Variable-order BDF with modified Newton and a reused finite-difference Jacobian (gl_oracle.c gl_oracle_bdf): the
    algorithm family and tolerances of the reference's CVODES call.  Returns (x_next, n_rhs_evals, [steps, jacobians, LUs, order])."""
    x, u, d, p = _c(x, NX), _c(u, NU), _c(d, ND), _c(p, NP)
    out = np.empty(NX)
    st = np.zeros(4)
    nfev = lib().gl_oracle_bdf(_p(x), _p(u), _p(d), _p(p), float(dt), float(rtol), float(atol), _p(out), _p(st))
    if nfev < 0:
        raise RuntimeError("gl_oracle_bdf: step size underflow")
    return out, int(nfev), st


def bdf_batch(X, U, D, p, dt=900.0, rtol=1e-6, atol=1e-6):
    """gl_oracle_bdf over the rows of X / U / D (shared p), inside one C call.  Returns (x_next [B, 28], total RHS evaluations)."""
    X, U, D, p = _c(X), _c(U), _c(D), _c(p, NP)
    B = X.shape[0]
    out = np.empty((B, NX))
    n = lib().gl_oracle_bdf_batch(_p(X), _p(U), _p(D), _p(p), B, float(dt), float(rtol), float(atol), _p(out))
    return out, int(n)


def scaled_rel_err(X, Xref):
    """max_t,i |X - Xref| / max(|Xref|, 1e-3 * max_t |Xref_i|)   (hard part 3)."""
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    Xref = np.atleast_2d(np.asarray(Xref, dtype=np.float64))
    scale = np.maximum(np.abs(Xref), 1e-3 * np.max(np.abs(Xref), axis=0, keepdims=True))
    scale = np.where(scale == 0.0, 1.0, scale)
    return float(np.max(np.abs(X - Xref) / scale))
