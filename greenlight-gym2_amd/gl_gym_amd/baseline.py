"""Rule-based greenhouse controller (config 1 of BASELINE.json: "fixed rule-based actions").

Behavioural mirror of ``RuleBasedController.predict`` (gl_gym/environments/baseline.py:68-227) with the constants of
gl_gym/configs/agents/rule_based.yml, vectorised over a batch of environments: every rule is a smooth
proportional band ``min + (max-min) / (1 + exp(-2/pBand * ln(100) * (v - setpoint - pBand/2)))``.

    u[B,6] = controller.predict(x[B,28], d[B,10], hour_of_day[B], day_of_year[B])

Works on numpy arrays or torch tensors (any device): only elementwise ops are used, so the batched config-1 loop
can stay on the GPU (PyTorch elementwise kernels are plumbing here; the hot path is glgym_step).
"""
from __future__ import annotations

import math

import numpy as np

RULE_BASED_DEFAULTS = dict(
    lamps_on=0, lamps_off=18, lamps_day_start=-1, lamps_day_stop=366, lamps_off_sun=400, lamp_rad_sum_limit=10,
    temp_setpoint_day=19.5, temp_setpoint_night=16.5, heat_correction=0, heat_deadzone=5, co2_day=800,
    vent_heat_Pband=4, rh_max=85, mech_dehumid_Pband=2, vent_rh_Pband=5, t_vent_off=1, vent_cold_Pband=-1,
    thScrSpDay=5, thScrSpNight=10, thScrPband=-1, thScrDeadZone=4, thScrRh=-2, thScrRhPband=2, lampExtraHeat=2,
    blScrExtraRh=100, rhMax=85, tHeatBand=-1, co2Band=-100, useBlScr=1)


class _Ops:
    """numpy / torch dispatch for the handful of elementwise ops the rules need."""

    def __init__(self, like):
        self.torch = None
        if type(like).__module__.startswith("torch"):
            import torch
            self.torch = torch

    def exp(self, v):
        return self.torch.exp(v) if self.torch else np.exp(v)

    def maximum(self, a, b):
        if self.torch:
            return self.torch.maximum(*self._both(a, b))
        return np.maximum(a, b)

    def minimum(self, a, b):
        if self.torch:
            return self.torch.minimum(*self._both(a, b))
        return np.minimum(a, b)

    def clip01(self, v):
        return self.torch.clamp(v, 0.0, 1.0) if self.torch else np.clip(v, 0.0, 1.0)

    def f(self, cond):      # boolean -> float
        return cond.to(self._dtype) if self.torch else cond.astype(np.float64)

    def _both(self, a, b):
        t = self.torch
        ref = a if t.is_tensor(a) else b
        return (a if t.is_tensor(a) else t.as_tensor(a, dtype=ref.dtype, device=ref.device),
                b if t.is_tensor(b) else t.as_tensor(b, dtype=ref.dtype, device=ref.device))

    def stack(self, cols):
        return self.torch.stack(cols, dim=-1) if self.torch else np.stack(cols, axis=-1)


class RuleBasedController:
    def __init__(self, **kw):
        cfg = dict(RULE_BASED_DEFAULTS, **kw)
        unknown = set(cfg) - set(RULE_BASED_DEFAULTS)
        if unknown:
            raise TypeError(f"unknown rule-based parameters: {sorted(unknown)}")
        self.__dict__.update(cfg)

    @staticmethod
    def _pband(ops, v, set_pt, p_band, lo, hi):
        return lo + (hi - lo) * (1.0 / (1.0 + ops.exp(-2.0 / p_band * math.log(100.0) * (v - set_pt - p_band / 2.0))))

    def predict(self, x, d, hour_of_day, day_of_year):
        ops = _Ops(x)
        if ops.torch:
            ops._dtype = x.dtype
        c = self
        hod, doy = hour_of_day, day_of_year
        tAir, vpAir, co2Air = x[..., 2], x[..., 15], x[..., 0]
        iGlob, tOut, dli, isDay, isDaySmooth = d[..., 0], d[..., 1], d[..., 7], d[..., 8], d[..., 9]
        pb = lambda v, sp, band, lo, hi: self._pband(ops, v, sp, band, lo, hi)  # noqa: E731

        # lamps by time of day / day of year (baseline.py:76-87)
        if c.lamps_on <= c.lamps_off:
            tod = ops.f((hod > c.lamps_on) & (hod < c.lamps_off))
        else:
            tod = ops.f((hod > c.lamps_on) | (hod < c.lamps_off))
        if c.lamps_day_start <= c.lamps_day_stop:
            doy_ok = ops.f((doy > c.lamps_day_start) & (doy < c.lamps_day_stop))
        else:
            doy_ok = ops.f((doy > c.lamps_day_start) | (doy < c.lamps_day_stop))
        below_dli = ops.f(dli < c.lamp_rad_sum_limit)
        lamp_no_cons = ops.f(iGlob < c.lamps_off_sun) * below_dli * tod * doy_ok                  # :98

        # linear one-hour ramps around switching times (:107-125)
        sw_on = ops.clip01(hod - c.lamps_on + 1)
        sw_off = ops.clip01(c.lamps_off - hod + 1)
        if c.lamps_on == c.lamps_off:
            both = sw_on * 0.0
        elif c.lamps_on < c.lamps_off:
            both = ops.minimum(sw_on, sw_off)
        else:
            both = ops.maximum(sw_on, sw_off)
        smooth_lamp = both * below_dli * doy_ok                                                    # :128
        is_day_inside = ops.maximum(smooth_lamp, isDay)                                            # :133

        heat_sp = is_day_inside * c.temp_setpoint_day + (1 - is_day_inside) * c.temp_setpoint_night \
            + c.heat_correction * lamp_no_cons                                                     # :136
        heat_max = heat_sp + c.heat_deadzone
        co2_sp = is_day_inside * c.co2_day
        co2_ppm = 1e6 * 8.3144598 * (tAir + 273.15) * (1e-6 * co2Air) / (101325 * 44.01e-3)        # :145
        rh_in = 100 * vpAir / (610.78 * ops.exp(17.2694 * tAir / (tAir + 238.3)))                  # :151

        vent_heat = pb(tAir, heat_max, c.vent_heat_Pband, 0, 1)
        vent_rh = pb(rh_in, c.rh_max + 0 * c.mech_dehumid_Pband, c.vent_rh_Pband, 0, 1)
        vent_cold = pb(tAir, heat_sp - c.t_vent_off, c.vent_cold_Pband, 1, 0)
        th_sp = isDay * c.thScrSpDay + (1 - isDay) * c.thScrSpNight
        th_cold = pb(tOut, th_sp, c.thScrPband, 0, 1)
        th_heat = pb(tAir, heat_sp + c.thScrDeadZone, -c.thScrPband, 1, 0)
        th_rh = ops.maximum(pb(rh_in, c.rhMax + c.thScrRh, c.thScrRhPband, 1, 0), 1 - vent_cold)
        lamp_on = lamp_no_cons * pb(tAir, heat_max + c.lampExtraHeat, -0.5, 0, 1) * (isDaySmooth + (1 - isDaySmooth)) \
            * ops.maximum(pb(rh_in, c.rhMax + c.blScrExtraRh, -0.5, 0, 1), 1 - vent_cold)         # :189-191

        return ops.stack([pb(tAir, heat_sp, c.tHeatBand, 0, 1),                                    # boiler
                          pb(co2_ppm, co2_sp, c.co2Band, 0, 1),                                    # CO2 dosing
                          ops.minimum(th_cold, ops.maximum(th_heat, th_rh)),                       # thermal screen
                          ops.minimum(vent_cold, ops.maximum(vent_heat, vent_rh)),                 # roof vents
                          lamp_on,                                                                 # lamps
                          c.useBlScr * (1 - isDaySmooth) * lamp_on])                               # blackout screen
