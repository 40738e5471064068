"""The 208-entry GreenLight parameter block, as a float32 vector (host side, built once per env).

Mirrors the *values and rounding behaviour* of the reference's ``init_default_params``
(gl_gym/environments/parameters.py:4-261): the block is a float32 array, derived entries
(108-126, 144, 169, 171) are computed from already-rounded float32 entries, and the result is
promoted to float64 when handed to the integrator (tomato_env.py:120 -> pybind ``std::vector<double>``).

Rounding of the derived entries depends on the NumPy generation the reference runs under:
its requirements.txt pins numpy 1.26 (float32-scalar x Python-float -> float64 intermediates, one
final rounding); under NumPy >= 2 (NEP 50) the same source computes them in float32.  ``semantics``
selects which one to reproduce; they differ by at most 1 float32 ulp in a handful of entries.
The committed fixture tests/golden/params_default.npz was produced under NumPy 2.2.
"""
from __future__ import annotations

import numpy as np

NP = 208

# index -> value, grouped by subsystem.  Entries not listed are derived below (or zero).
_PHYSICS = {0: 5., 1: 2.45e6, 2: 5.67e-8, 3: 1., 4: 1., 5: 0.5, 6: 0.5, 7: 0.554, 8: 0.9, 9: 1.2, 10: 0.07, 11: 0.35,
            12: 7850., 13: 1000., 14: 65.8, 15: 1.99e-7, 16: 1200., 17: 4.3, 18: 0.54, 19: 6.1e-7, 20: 1.1e-11,
            21: 4.3e-6, 22: 5.2e-6, 23: 1000., 24: 640., 25: 4180., 26: 9.81}
_SOIL_CANOPY = {27: 0.04, 28: 0.08, 29: 0.16, 30: 0.32, 31: 0.64, 32: 0.7, 33: 0.7, 34: 0.27, 35: 0.94, 36: 28.96,
                37: 1.28, 38: 18., 39: 8314., 40: 5., 41: 275., 42: 82., 43: -1.}
_CONSTRUCTION = {44: 0.1, 45: 23., 46: 144., 47: 216.6, 48: 5.7, 49: 6.2, 50: 3.5, 51: 2.8, 52: 1.2, 53: 1., 54: 0.,
                 55: 52.2, 56: 0.87, 57: 1., 58: 0., 59: 0.35, 60: 0.3e-4, 61: 0.02, 62: 0.}
_ROOF = {63: 0.85, 64: 2600., 65: 0.13, 66: 0.13, 67: 0.15, 68: 0.57, 69: 0.57, 70: 0., 71: 1.05, 72: 840., 73: 4e-3}
_THERMAL_SCREEN = {74: 0.67, 75: 200., 76: 0.35, 77: 0.35, 78: 0.18, 79: 0.75, 80: 0.75, 81: 0.15, 82: 1800.,
                   83: 0.35e-3, 84: 5.e-4}
_BLACKOUT_SCREEN = {85: 0.67, 86: 200., 87: 0.35, 88: 0.35, 89: 0.01, 90: 0.01, 91: 0.7, 92: 1800., 93: 0.35e-3,
                    94: 5.e-4}
_FLOOR_SOIL = {95: 1., 96: 2300., 97: 0.5, 98: 0.65, 99: 1.7, 100: 880., 101: 0.02, 102: 1_730_000., 103: 0.85}
_PIPES = {104: 0.88, 105: 51.e-3, 106: (51.e-3) - (2.25e-3), 107: 1.3375}
_CROP = {127: 31.65, 128: 2.3, 129: 210., 130: 1.7, 131: 0.67, 132: 37000, 133: 298.15, 134: 710, 135: 220_000,
         136: 0.7, 137: 0.385, 138: 30e-3, 139: 44e-3, 140: 4.6, 141: 3.0, 142: 2.66e-5, 143: 3e-6, 145: 3_000_000,
         146: 0.27, 147: 0.28, 148: 0.3, 149: 2_850_000, 150: 2., 151: 1.16e-7, 152: 3.47e-7, 153: 1.47e-7,
         154: 0.328, 155: 0.095, 156: 0.074, 157: 20e3, 158: 1e3, 159: 24.5, 160: 15, 161: 34, 162: 10, 163: 1035,
         164: 1250}
_GROW_PIPES = {165: 0, 166: 1.655, 167: 35e-3, 168: (35e-3) - (1.2e-3), 170: 0}
_LAMPS = {172: 116, 173: 0, 174: 0.31, 175: 0.02, 176: 0.95, 177: 0.95, 178: 0.95, 179: 0., 180: 0., 181: 0.05,
          182: 0.88, 183: 0.88, 184: 10., 185: 2.3, 186: 0.63, 187: 5.2}
_INTERLIGHTS = {188: 0, 189: 0.5, 190: 0.5, 191: 10, 192: 0, 193: 0, 194: 0, 195: 0, 196: 0, 197: 0, 198: 0, 199: 1,
                200: 1.4, 201: 1.4, 202: 0.54, 203: 1.88}
_MISC = {204: 0.9, 205: 0.25, 206: 0.0627, 207: 1e-6}


def init_default_params(nparams: int = NP, semantics: str = "numpy2") -> np.ndarray:
    """float32[208].  semantics: "numpy2" (float32 intermediates) or "numpy1" (float64 intermediates)."""
    if nparams != NP:
        raise ValueError("the GreenLight model has 208 parameters")
    if semantics not in ("numpy1", "numpy2"):
        raise ValueError(semantics)
    p = np.zeros(NP, dtype=np.float32)
    for group in (_PHYSICS, _SOIL_CANOPY, _CONSTRUCTION, _ROOF, _THERMAL_SCREEN, _BLACKOUT_SCREEN, _FLOOR_SOIL,
                  _PIPES, _CROP, _GROW_PIPES, _LAMPS, _INTERLIGHTS, _MISC):
        for i, v in group.items():
            p[i] = v

    # In "numpy2" mode g(i) is a float32 scalar and Python floats are weak, so each operation rounds to
    # float32; in "numpy1" mode everything is float64 until the store.
    g = (lambda i: p[i]) if semantics == "numpy2" else (lambda i: float(p[i]))
    pi = np.pi

    def pipe_capacity(length, d_ext, d_int):    # steel wall + water filling, per floor area
        return 0.25 * pi * g(length) * ((g(d_ext) * g(d_ext) - g(d_int) * g(d_int)) * g(12) * g(24)
                                        + g(d_int) * g(d_int) * g(13) * g(25))

    p[108] = 130. * g(46)                                  # boiler capacity [W]
    p[109] = 5.0 * g(46)                                   # CO2 supply capacity [mg s-1]
    p[110] = pipe_capacity(107, 105, 106)
    p[111] = g(9) * np.exp(g(26) * g(36) * g(54) / (g(39) * 293.15))       # air density at altitude
    p[112] = g(48) * g(111) * g(23)                        # heat capacities
    p[113] = g(101) * g(96) * g(100)
    for k, layer in enumerate((27, 28, 29, 30, 31)):
        p[114 + k] = g(layer) * g(102)
    p[119] = g(83) * g(75) * g(82)
    p[120] = (g(49) - g(48)) * g(111) * g(23)
    p[121] = g(93) * g(86) * g(92)
    p[122] = g(48)                                         # CO2 capacities [m]
    p[123] = g(49) - g(48)
    p[124] = pi * g(107) * g(105)                          # pipe surface per floor area
    p[125] = 1 - 0.49 * pi * g(107) * g(105)               # canopy -> floor view factor
    p[126] = 101325 * pow((1 - 2.5577e-5 * g(54)), 5.25588)
    p[144] = g(141) / g(142)                               # cLeafMax = laiMax / sla
    p[169] = pi * g(166) * g(167)
    p[171] = pipe_capacity(166, 167, 168)
    return p
