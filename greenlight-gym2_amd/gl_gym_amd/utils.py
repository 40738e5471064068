"""Host-side helpers around the hot path: initial state, unit conversions, weather tensors.

Behavioural mirror of gl_gym/environments/utils.py (init_state :13-46, load_weather_data :48-125,
computeisDay :177-214, dailLightSum :216-250, soilTempNl :262-279, unit conversions :281-444).
These run once per reset on the host and produce the ``[T,10]`` disturbance tensor that is uploaded to HBM;
the per-step work happens in the HIP kernels.
"""
from __future__ import annotations

from os.path import join

import numpy as np

SECS_PER_DAY = 86400
_R, _C2K, _M_CO2, _M_H2O, _P_ATM = 8.3144598, 273.15, 44.01e-3, 18.01528e-3, 101325


# ---- unit conversions -------------------------------------------------------------------------
def satVp(temp):
    return 610.78 * np.exp(17.2694 * temp / (temp + 238.3))


def co2ppm2dens(temp, ppm):
    return _P_ATM * 10 ** -6 * ppm * _M_CO2 / (_R * (temp + _C2K))


def co2dens2ppm(temp, dens):
    return 1e6 * _R * (temp + _C2K) * dens / (_P_ATM * _M_CO2)


def rh2vaporDens(temp, rh):
    return (rh / 100) * satVp(temp) * _M_H2O / (_R * (temp + _C2K))


def vaporDens2pres(temp, vaporDens):
    return satVp(temp) * (vaporDens / rh2vaporDens(temp, 100))


def vaporPres2rh(temp, vaporPres):
    return np.clip(100 * vaporPres / satVp(temp), a_min=0., a_max=100.)


def vaporDens2rh(temp, vaporDens):
    """Vapour density [kg m-3] -> relative humidity [%], clipped to 0..100 (ideal gas law over the Magnus saturation pressure)."""
    return np.clip(100.0 * 8.3144598 * (temp + 273.15) / (18.01528e-3 * satVp(temp)) * vaporDens, 0, 100)


def compute_sky_temp(air_temp, cloud):
    """Sky temperature [C] from air temperature [C] and cloud cover (0-1): clear-sky long-wave 213 + 5.5 T, emissivity
    blended towards 1 with 0.84 * cloud, radiating temperature of the result."""
    sigma, c2k = 5.67e-8, 273.15
    t4 = sigma * (air_temp + c2k) ** 4
    eps = (1 - 0.84 * cloud) * (213 + 5.5 * air_temp) / t4 + 0.84 * cloud
    return (eps * t4 / sigma) ** 0.25 - c2k


def days2date(timeInDays, referenceDate):
    """Days since ``referenceDate`` ('DD-MM-YYYY') -> list of 'YYYY-MM-DD HH:MM:SS' strings (minutes truncated)."""
    from datetime import datetime, timedelta
    ref = datetime.strptime(referenceDate, "%d-%m-%Y")
    t = np.atleast_1d(np.asarray(timeInDays, dtype=np.float64))
    days = np.floor(t).astype(int)
    hours_f = (t - days) * 24
    hours = hours_f.astype(int)
    minutes = ((hours_f - hours) * 60).astype(int)
    return [(ref + timedelta(days=int(d), hours=int(h), minutes=int(m))).strftime("%Y-%m-%d %H:%M:%S")
            for d, h, m in zip(days, hours, minutes)]


def soilTempNl(time):
    year = 3600 * 24 * 365
    return 10 + 5 * np.sin((2 * np.pi * (time + 0.625 * year) / year))


# ---- initial state ----------------------------------------------------------------------------
def init_state(d0, rhMax=90, time_in_days=0):
    """x0[28] from the first weather row (mature crop, 16.5 C greenhouse)."""
    t_air, t_so_out = 16.5, d0[6]
    x = np.full(28, t_air)
    x[0] = x[1] = d0[3]
    x[4] = x[21] = t_air + 4
    x[11:15] = [0.25 * (3. * t_air + t_so_out), 0.25 * (2. * t_air + 2 * t_so_out), 0.25 * (t_air + 3 * t_so_out),
                t_so_out]
    x[15] = x[16] = rhMax / 100. * satVp(t_air)
    x[22:28] = [0., 9.5283e4, 2.5107e5, 5.5338e4, 3.0978e3, time_in_days]
    return x


# ---- weather ----------------------------------------------------------------------------------
def daily_light_sum(time, rad, c=SECS_PER_DAY):
    """DLI [MJ m-2 day-1] per sample: the radiation sum of the calendar day the sample lies in.
    Reproduces the reference's segment boundaries, including its asymmetric midnight search."""
    interval = time[1] - time[0]
    day = np.floor(time / c)
    jumps = np.where(np.diff(day) == 1)[0]
    n = len(time)
    out = np.zeros(n)
    before = 0
    after = int(jumps[0] + 1) if jumps.size else n
    i = 0
    while i < n:
        seg_end = min(after, n)                       # samples i .. after-1 share one sum
        out[i:seg_end] = np.sum(rad[before:after + 1])
        i = seg_end
        if i >= n:
            break
        before = after
        nxt = jumps[jumps >= before + 2]              # later searches start two samples on and keep the raw index
        after = int(nxt[0]) if nxt.size else n
    return out * interval * 1e-6


def compute_is_day(rad, dt):
    """(isDay, isDaySmooth): 0/1 daylight flags with a one-hour linear / sigmoid ramp at sunrise and sunset."""
    is_day = (rad > 0) * 1.0
    smooth = is_day.copy()
    n_tr = int(3600 / dt)
    ramp = np.linspace(0, 1, n_tr)
    ramp_s = 1 / (1 + np.exp(-10 * (ramp - 0.5)))
    half = n_tr // 2
    in_sunset = False
    for k in range(n_tr, len(is_day) - n_tr):          # sequential: earlier ramps are visible to later tests
        cur, nxt = is_day[k], is_day[k + 1]
        if cur == 0:
            in_sunset = False
            if nxt == 1:
                is_day[k - half:k + half] = ramp
                smooth[k - half:k + half] = ramp_s
        elif cur == 1 and nxt == 0 and not in_sunset:
            is_day[k - half:k + half] = 1 - ramp
            smooth[k - half:k + half] = 1 - ramp_s
            in_sunset = True
    return is_day, smooth


# the reference's spellings of the two helpers above (gl_gym/environments/utils.py:177, 214)
dailLightSum = daily_light_sum
computeisDay = compute_is_day


def expandWeatherData(weatherDataDir, rawWeather, location, source, growthYear, time, dt):
    """Append next year's CSV (time column shifted to continue after ``time[-1]``) to ``rawWeather``."""
    import pandas as pd
    nxt = pd.read_csv(join(join(weatherDataDir, location), source + str(growthYear + 1)) + ".csv", sep=",")
    nxt["time"] += time[-1] + dt
    return pd.concat([rawWeather, nxt.iloc[:, :]])


def weather_from_raw(time, i_glob, t_out, rh, wind, t_sky, h, nd=10, co2_ppm=400):
    """Raw (already sliced) columns -> resampled [ns, nd] disturbance tensor."""
    from scipy.interpolate import PchipInterpolator
    time = np.asarray(time, dtype=np.float64)
    dt = np.mean(np.diff(time - time[0]))
    w = np.zeros((len(time), nd))
    w[:, 0] = i_glob
    w[:, 1] = t_out
    w[:, 2] = vaporDens2pres(w[:, 1], rh2vaporDens(w[:, 1], np.asarray(rh, dtype=np.float64)))
    w[:, 3] = co2ppm2dens(w[:, 1], co2_ppm) * 1e6
    w[:, 4] = wind
    w[:, 5] = t_sky
    w[:, 6] = soilTempNl(time)
    w[:, 7] = daily_light_sum(time, w[:, 0])
    w[:, 8], w[:, 9] = compute_is_day(w[:, 0], dt)
    ns = int((dt / h) * len(time))
    out = PchipInterpolator(time, w)(np.linspace(time[0], time[-1], ns))
    out[:, 0][out[:, 0] < 1e-10] = 0
    return out


def load_weather_data(weatherDataDir, location, source, growthYear, startDay, nDays, predHorizon, h, nd):
    """CSV -> [ns, nd] tensor, same call signature as the reference (predHorizon is in DAYS there)."""
    import pandas as pd
    path = join(join(weatherDataDir, location), source + str(growthYear)) + ".csv"
    raw = pd.read_csv(path, sep=",")
    time = raw["time"].values
    dt = np.mean(np.diff(time - time[0]))
    n0 = int(np.ceil(startDay * SECS_PER_DAY / dt))
    n_tot = int(np.ceil(nDays * SECS_PER_DAY / dt)) + int(np.ceil(predHorizon * SECS_PER_DAY / dt)) + 1
    if n0 + n_tot > len(time):                           # season runs into the next year's file
        nxt = pd.read_csv(join(join(weatherDataDir, location), source + str(growthYear + 1)) + ".csv", sep=",")
        nxt["time"] += time[-1] + dt
        raw = pd.concat([raw, nxt.iloc[:, :]])
    sl = slice(n0, n0 + n_tot)
    return weather_from_raw(raw["time"].values[sl], raw["global radiation"].values[sl],
                            raw["air temperature"].values[sl], raw["RH"].values[sl], raw["wind speed"].values[sl],
                            raw["sky temperature"].values[sl], h, nd)


def synthetic_weather(n_rows=35040, dt=900.0, seed=2024):
    """Synthetic one-year [n_rows, 10] disturbance tensor: used by bench.py and the tests
    in place of the Amsterdam KNMI files, which are not redistributable with this repository."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_rows) * dt
    hour = (t / 3600.0) % 24.0
    i_glob = np.maximum(0.0, 700.0 * np.sin(np.pi * (hour - 6.0) / 12.0)) * rng.uniform(0.3, 1.0, n_rows)
    i_glob[(hour < 6.0) | (hour > 18.0)] = 0.0
    t_out = 10.0 + 6.0 * np.sin(2 * np.pi * (hour - 9.0) / 24.0) + rng.standard_normal(n_rows)
    rh = np.clip(80.0 - 1.5 * (t_out - 10.0) + 5.0 * rng.standard_normal(n_rows), 30.0, 100.0)
    w = np.zeros((n_rows, 10))
    w[:, 0] = i_glob
    w[:, 1] = t_out
    w[:, 2] = vaporDens2pres(t_out, rh2vaporDens(t_out, rh))
    w[:, 3] = co2ppm2dens(t_out, 400.0) * 1e6
    w[:, 4] = rng.lognormal(np.log(3.5), 0.5, n_rows)
    w[:, 5] = t_out - rng.uniform(5.0, 20.0, n_rows)
    w[:, 6] = soilTempNl(t)
    w[:, 7] = daily_light_sum(t, i_glob)
    w[:, 8], w[:, 9] = compute_is_day(i_glob, dt)
    return w
