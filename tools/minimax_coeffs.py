#!/usr/bin/env python3
"""Near-minimax (Chebyshev-interpolation) polynomial coefficients for the lean fp64 exp / log of gl_model.hpp, computed in
80-bit long double, and their measured worst relative error when evaluated in fp64 by Horner with FMA-free arithmetic
(an upper bound for the device's FMA chain).      python tools/minimax_coeffs.py"""
import numpy as np
L = np.longdouble
PI = L("3.14159265358979323846264338327950288")


def cheb_fit(f, a, b, deg):
    """monomial coefficients (in t = x, low order first, long double) of the degree-`deg` Chebyshev interpolant of f on [a, b]"""
    n = deg + 1
    j = np.arange(n, dtype=L)
    tk = np.cos(PI * (j + L(0.5)) / n)                    # nodes on [-1, 1]
    x = (a + b) / 2 + (b - a) / 2 * tk
    y = f(x)
    c = np.array([(L(2) / n) * np.sum(y * np.cos(PI * k * (j + L(0.5)) / n)) for k in range(n)], dtype=L)
    c[0] /= 2
    # Chebyshev series in t -> monomials in t (recurrence), then t = (2 x - a - b) / (b - a) -> monomials in x
    T = [np.array([1], dtype=L), np.array([0, 1], dtype=L)]
    for k in range(2, n):
        nxt = np.zeros(k + 1, dtype=L); nxt[1:] += 2 * T[-1]; nxt[:k - 1] -= T[-2]
        T.append(nxt)
    mono_t = np.zeros(n, dtype=L)
    for k in range(n):
        mono_t[:k + 1] += c[k] * T[k]
    al, be = 2 / (b - a), -(a + b) / (b - a)                    # t = al x + be
    out = np.zeros(n, dtype=L)
    p = np.array([1], dtype=L)                                   # (al x + be)^k
    for k in range(n):
        out[:k + 1] += mono_t[k] * p
        nxt = np.zeros(len(p) + 1, dtype=L); nxt[:-1] += be * p; nxt[1:] += al * p; p = nxt      # np.convolve drops to double
    return out


def horner64(c, x):
    r = np.full_like(x, float(c[-1]))
    for v in c[-2::-1]:
        r = r * x + float(v)
    return r


if __name__ == "__main__":
    a = L(np.log(L(2))) / 2 * L(1.0001)
    xs = np.linspace(-float(a), float(a), 400001)
    for deg in (10, 11, 12, 13):
        c = cheb_fit(lambda x: np.exp(x), -a, a, deg)
        err = np.max(np.abs(horner64(c, xs).astype(L) / np.exp(xs.astype(L)) - 1))
        r = np.full(len(xs), c[-1], dtype=L)
        for v in c[-2::-1]:
            r = r * xs.astype(L) + v
        print("exp degree", deg, "max rel err: truncation %.2e, fp64 Horner without FMA %.2e" % (float(np.max(np.abs(r / np.exp(xs.astype(L)) - 1))), float(err)))
        if deg in (10, 11):
            print("   ", ", ".join("%.17e" % float(v) for v in c))
    # log: ln m = 2 s (1 + z q(z)), s = (m - 1) / (m + 1), z = s^2 in [0, zb], q(z) = (atanh(s) / s - 1) / z = 1/3 + z/5 + ...
    sb = (L(2).sqrt() - 1) / (L(2).sqrt() + 1) * L(1.0001) if hasattr(L(2), "sqrt") else L((np.sqrt(L(2)) - 1) / (np.sqrt(L(2)) + 1)) * L(1.0001)
    zb = sb * sb
    def q(z):
        s = np.sqrt(z)
        out = np.empty_like(z)
        small = z < 1e-6
        out[small] = 1 / L(3) + z[small] / 5 + z[small] ** 2 / 7
        zz, ss = z[~small], s[~small]
        # atanh via log1p in long double
        out[~small] = ((np.log1p(2 * ss / (1 - ss)) / 2) / ss - 1) / zz
        return out
    zs = np.linspace(0.0, float(zb), 400001)
    ss = np.sqrt(zs.astype(L))
    true = np.where(ss > 0, np.log1p(2 * ss / (1 - ss)) / 2, 0)
    for deg in (5, 6, 7, 9):
        c = cheb_fit(q, L(0), zb, deg)
        approx = ss * (1 + zs.astype(L) * horner64(c, zs).astype(L))
        m = ss > 1e-3
        err = np.max(np.abs(approx[m] / true[m] - 1))
        print("log q degree", deg, "max rel err of atanh(s) %.2e" % float(err))
        if deg in (5, 6, 7):
            print("   ", ", ".join("%.17e" % float(v) for v in c))
