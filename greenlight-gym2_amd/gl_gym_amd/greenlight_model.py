"""Drop-in for the reference's pybind11 module ``gl_gym.environments.models.greenlight_model``
(greenlight_model.cpp:130-136): same class name, constructor signature and ``evalF`` contract, but the
step map runs on the MI355X through libglgym.so (sub-stepped RK4 instead of CasADi/CVODES).

    GreenLight(nx, nu, nd, np, dt).evalF(x, u, d, p) -> list[float]   # len 28

Differences a maintainer should know about: errors surface as ``GlgymError`` (a RuntimeError, like the
pybind-translated CasADi exceptions); no JIT artefacts are written to the CWD; the destructor is silent.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .parameters import init_default_params


class GreenLight:
    def __init__(self, nx, nu, nd, np_, dt, dtype="float64", n_sub=None, device=0, variant="ode", scheme=None, window=None,
                 preset="parity"):
        """nd = 10, or 14 as in experiments/gl_predefined_controls.py:95 (rows carry the measured pipe columns).
        variant = "ode" (what the reference's compiled module integrates) or "ode_pipe" (ode.hpp:126-263, nd >= 14).
        scheme = "ls5" (default; five-stage fourth-order 2N scheme), "rk4", "rk3" or "rk2" (include/glgym.h); variant "ode_pipe"
        defaults to "rk4" and accepts no other scheme.
        preset = "parity" (default HERE: this class stands in for the reference's CVODES call, so it integrates inside the band that
        solver's tolerances keep from the tight solution -- ls5: n_sub 192, one sub-step per window, 9.1e-6 on the tight one-step
        tuples) or "throughput" (n_sub 128, window 2: 5.4e-5; what the batched envs run); n_sub / window override the preset."""
        self._lib = L.load()
        scheme = L.resolve_scheme(scheme, variant)           # None: "ls5"; "rk4" for ode_pipe (the only scheme its kernels are built for)
        if preset not in L.PRESETS:
            raise ValueError("preset must be 'parity' or 'throughput'")
        n_def, w_def = L.preset_n_sub(scheme, dt, preset)
        window = (w_def if n_sub is None else 0) if window is None else window
        n_sub = n_def if n_sub is None else n_sub
        self.scheme, self.n_sub, self.window, self.preset = scheme, int(n_sub), int(window), preset
        self.nx, self.nu, self.nd, self.np = int(nx), int(nu), int(nd), int(np_)
        self.dt = float(dt)
        self._h = C.c_void_p()
        p = np.ascontiguousarray(init_default_params(L.NP), dtype=np.float64) if self.np == L.NP else np.zeros(1)
        rc = self._lib.glgym_create(self.nx, self.nu, self.nd, self.np, self.dt, p.ctypes.data_as(L._DP),
                                    L.F64 if str(dtype) in ("float64", "f64", "double") else L.F32, int(n_sub),
                                    int(device), C.byref(self._h))
        L.check(rc, "glgym_create")
        if variant == "ode_pipe":
            L.check(self._lib.glgym_set_model_variant(self._h, L.ODE_PIPE), "glgym_set_model_variant")
        L.check(self._lib.glgym_set_scheme(self._h, L.SCHEMES[scheme]), "glgym_set_scheme")
        L.check(self._lib.glgym_set_window(self._h, self.window), "glgym_set_window")

    @property
    def handle(self):
        return self._h

    def set_n_sub(self, n_sub):
        L.check(self._lib.glgym_set_n_sub(self._h, int(n_sub)), "glgym_set_n_sub")
        self.n_sub = int(n_sub)

    def set_window(self, window):
        L.check(self._lib.glgym_set_window(self._h, int(window)), "glgym_set_window")
        self.window = int(window)

    def set_verify(self, mode: str):
        """Step-doubling verified integration: "auto" (default; evalF is always verified: it takes any u), "always", "never"
        (include/glgym.h, glgym_verify)."""
        L.check(self._lib.glgym_set_verify(self._h, L.VERIFY_MODES[mode]), "glgym_set_verify")

    def set_ladder_parallel(self, on: bool):
        """Verified evalF calls on small batches run two rungs of the step-doubling ladder side by side (default; include/glgym.h
        glgym_set_ladder_parallel): identical results, two thirds of the latency.  False: always the sequential ladder."""
        L.check(self._lib.glgym_set_ladder_parallel(self._h, int(bool(on))), "glgym_set_ladder_parallel")

    def set_layout(self, layout: str):
        """Kernel layout of fp32 handles ("auto" | "one" | "quad"; include/glgym.h glgym_set_layout): evalF rows without their own
        parameter block run four lanes per row up to 16 384 rows, one lane per row beyond; fp64 has one layout."""
        L.check(self._lib.glgym_set_layout(self._h, L.LAYOUTS[layout]), "glgym_set_layout")

    def _as(self, a, n, B):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(B, -1)
        if a.shape[1] != n:
            raise ValueError(f"expected {n} columns, got {a.shape[1]}")
        return a

    def evalF(self, x, u, d, p):
        """x(dt) for x' = ODE(x; u, d, p), (u, d, p) held over the step.  Returns a Python list like the reference."""
        return self.evalF_batch(x, u, d, p)[0].tolist()

    def evalF_batch(self, x, u, d, p=None):
        """Row-major batch version: x[B,28], u[B,6], d[B,nd], p[208] or p[B,208] -> ndarray [B,28]."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = 1 if x.ndim == 1 else x.shape[0]
        x, u, d = self._as(x, L.NX, B), self._as(u, L.NU, B), self._as(d, self.nd, B)
        out = np.empty((B, L.NX))
        if p is None:
            pp, rows = None, 1
        else:
            p = np.ascontiguousarray(p, dtype=np.float64)
            rows = 1 if p.ndim == 1 else p.shape[0]
            p = p.reshape(rows, L.NP)
            if rows not in (1, B):
                raise ValueError("p must be [208] or [B,208]")
            pp = p.ctypes.data_as(L._DP)
        rc = self._lib.glgym_evalF(self._h, x.ctypes.data_as(L._DP), u.ctypes.data_as(L._DP), d.ctypes.data_as(L._DP),
                                   pp, rows, B, out.ctypes.data_as(L._DP))
        L.check(rc, "glgym_evalF")
        return out

    def rhs(self, x, u, d):
        """dx/dt with the handle's parameter block (test hook)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        B = 1 if x.ndim == 1 else x.shape[0]
        x, u, d = self._as(x, L.NX, B), self._as(u, L.NU, B), self._as(d, self.nd, B)
        out = np.empty((B, L.NX))
        L.check(self._lib.glgym_rhs(self._h, x.ctypes.data_as(L._DP), u.ctypes.data_as(L._DP),
                                    d.ctypes.data_as(L._DP), B, out.ctypes.data_as(L._DP)), "glgym_rhs")
        return out

    def set_params(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        L.check(self._lib.glgym_set_params(self._h, p.ctypes.data_as(L._DP)), "glgym_set_params")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.glgym_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
