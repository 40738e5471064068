"""ctypes binding of libglgym.so (include/glgym.h).  No CPU fallback: a missing library or a missing
HIP device raises -- the product path never routes through the oracle or any host implementation."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("GLGYM_LIB", _HERE / "libglgym.so"))

NX, NU, ND, NP, NCROP, NINFO, NMETRIC = 28, 6, 10, 208, 34, 11, 14
METRIC_REPLICAS, METRIC_STRIDE = 64, 32          # glgym.h: the metric accumulators are replicated per cache line
F32, F64 = 0, 1
ODE, ODE_PIPE = 0, 1
SCHEME_RK4, SCHEME_RK2 = 0, 1
SCHEME_RK3 = 2
SCHEME_LS5 = 3
SCHEMES = {"rk4": SCHEME_RK4, "rk2": SCHEME_RK2, "rk3": SCHEME_RK3, "ls5": SCHEME_LS5}
ABI_VERSION = 6                                           # include/glgym.h GLGYM_ABI_VERSION
LAYOUTS = {"auto": 0, "one": 1, "quad": 2}                # glgym_layout
# NOMINAL sub-steps per 900 s env-step.  RK4 (round 4): the conduction between the two faces of the cover glass -- the 0.65 1/s
# mode that kept every explicit scheme at >= 224 sub-steps -- is integrated exactly (gl_model.hpp rk_delta, COVEXP), so the nominal
# sub-step is set by the top compartment's air exchange: 240 covers rates up to 0.68 1/s (exceeded in 2e-5 of random-action
# env-steps on the synthetic weather year; median 0.19, 99.9 % 0.55).  The kernels are stability-controlled per environment: an
# environment whose rate bound at the start of the env-step asks for more gets proportionally more windows (its own sub-step
# length), and what changes inside the env-step is followed window by window.  "rk3" is the three-stage member of the same
# exponential family (stability interval 2.513: 270 covers 0.69 1/s), "rk2" its midpoint rule (2.0: 336 covers 0.69 1/s).
# "ls5" (round 5, the default): the five-stage fourth-order 2N-storage scheme (stability interval 5.009: 1.00 per right-hand side
# against RK4's 0.70; cover conduction exact as in the others): 128 covers 0.655 1/s, two sub-steps per window (14 s; RK4-240: 15 s)
# -- 640 right-hand sides per env-step for RK4-240's 960 at the same accuracy on every fixture.
DEFAULT_N_SUB = {"rk4": 240, "rk2": 336, "rk3": 270, "ls5": 128}
DEFAULT_SCHEME = "ls5"
VERIFY_MODES = {"auto": 0, "always": 1, "never": 2}     # glgym_verify (include/glgym.h)
N_SUB_MULTIPLE = {"rk4": 4, "rk2": 4, "rk3": 3, "ls5": 2}          # tier-2b window of the scheme
# PRESETS.  "throughput": the scheme's nominal count with its own window (max scaled error on the tight one-step tuples 5.4e-5 for
# ls5, 6.1e-5 for rk4, 10-day rollout 1.5e-5 fp64 / 2.9e-5 fp32; bar 1e-4).  "parity": inside the 1.3e-5 band a BDF solve at the reference's
# tolerances (greenlight_model.cpp:51-52) keeps from the tight solution -- ls5: n_sub 192 with ONE sub-step per window (1.0e-5), the
# others: n_sub x 8/3 with their own window (rk4 640: 8.7e-6).  (n_sub, window) at dt = 900 s; window 0 = the scheme's own.
PRESETS = {"throughput": {k: (v, 0) for k, v in DEFAULT_N_SUB.items()},
           "parity": {"ls5": (192, 1), "rk4": (640, 0), "rk3": (720, 0), "rk2": (896, 0)}}


def preset_n_sub(scheme: str, dt: float, preset: str = "throughput"):
    """-> (n_sub, window) of `preset` for `scheme`, n_sub scaled with dt so that the nominal sub-step h = dt / n_sub stays the same
    (e.g. ls5: 7.03 s, 44 sub-steps at the dt = 300 s of experiments/run_time.py) and rounded up to a multiple of the window."""
    n0, window = PRESETS[preset][scheme]
    n = n0 * float(dt) / 900.0
    mult = window if window > 0 else N_SUB_MULTIPLE[scheme]
    return max(mult, int(-(-n // mult) * mult)), window


def resolve_scheme(scheme, variant: str = "ode") -> str:
    """The scheme a constructor ends up with.  None = the default: "ls5" for the default ODE; "rk4" for variant "ode_pipe", whose
    kernels are instantiated for GLGYM_SCHEME_RK4 only (glgym_step / glgym_evalF return GLGYM_EINVAL for any other scheme with
    GLGYM_ODE_PIPE) -- so TomatoVecEnv(model_variant="ode_pipe") and GreenLight(variant="ode_pipe") work with default arguments.
    An explicit other scheme with ode_pipe is refused here, by name, instead of at the first step."""
    if variant not in ("ode", "ode_pipe"):
        raise ValueError("variant must be 'ode' or 'ode_pipe'")
    if scheme is None:
        return "rk4" if variant == "ode_pipe" else DEFAULT_SCHEME
    if scheme not in SCHEMES:
        raise ValueError("scheme must be 'ls5', 'rk4', 'rk3' or 'rk2'")
    if variant == "ode_pipe" and scheme != "rk4":
        raise ValueError(f"variant 'ode_pipe' is built for scheme 'rk4' only (got {scheme!r}): leave scheme unset or pass scheme='rk4'")
    return scheme


def default_n_sub(scheme: str, dt: float) -> int:
    """Nominal sub-steps per env-step of the throughput preset (see preset_n_sub)."""
    return preset_n_sub(scheme, dt, "throughput")[0]
OK, EINVAL, ENODEV, EHIP, ENOMEM, EODE = 0, -1, -2, -3, -4, -5

INFO_KEYS = ("EPI", "revenue", "variable_costs", "fixed_costs", "co2_cost", "heat_cost", "elec_cost",
             "temp_violation", "co2_violation", "rh_violation", "lamp_violation")      # tomato_env.py:208-222
METRIC_KEYS = ("sum_reward", "sum_EPI", "n_done", "n_ode_fail", "sum_co2_violation", "sum_temp_violation",
               "sum_rh_violation", "n_env_steps", "n_guard_retries", "n_refined_substeps", "n_flag_err", "n_flag_branch",
               "n_flag_cap", "n_flag_heavy")


class GlgymError(RuntimeError):
    pass


class GlgymOdeError(GlgymError):
    """glgym_evalF: the integration failed for at least one row (GLGYM_EODE) -- the counterpart of the RuntimeError the
    reference's evalF raises when CVODES fails (greenlight_model.cpp:110, caught at tomato_env.py:119-123)."""


class RewardCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "elec_price", "heating_price", "co2_price", "fruit_price", "dmfm", "fixed_greenhouse_cost", "fixed_co2_cost",
        "fixed_lamp_cost", "fixed_screen_cost", "pen_lamp", "co2_min", "co2_max", "temp_min", "temp_max", "rh_min",
        "rh_max")]


class StepArgs(C.Structure):
    # struct_size first (since ABI 5): glgym_step refuses a struct of another size; make_step_args() fills it
    _fields_ = [("struct_size", C.c_int32), ("B", C.c_int32), ("ld", C.c_int32), ("x", C.c_void_p), ("u", C.c_void_p), ("action", C.c_void_p),
                ("control", C.c_void_p), ("weather", C.c_void_p), ("weather_rows", C.c_int32), ("w_off", C.c_void_p),
                ("timestep", C.c_void_p), ("crop_p", C.c_void_p), ("N", C.c_int32), ("reward", C.c_void_p),
                ("info", C.c_void_p), ("done", C.c_void_p), ("metrics", C.c_void_p), ("step_flags", C.c_void_p)]


def make_step_args(*args, **kw):
    """StepArgs(...) without the leading struct_size, which is filled in here."""
    return StepArgs(C.sizeof(StepArgs), *args, **kw)


# step_flags bits (include/glgym.h GLGYM_SF_*)
SF_FIRST_MASK, SF_ACCEPT_AGREE_FLAGGED, SF_ACCEPT_LAST_ALONE, SF_FAILED = 31, 32, 64, 128


class ObsArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("ld", C.c_int32), ("x", C.c_void_p), ("u", C.c_void_p), ("weather", C.c_void_p),
                ("weather_rows", C.c_int32), ("w_off", C.c_void_p), ("timestep", C.c_void_p), ("start_day", C.c_void_p),
                ("Np", C.c_int32), ("obs", C.c_void_p), ("mask", C.c_void_p), ("term_obs", C.c_void_p)]


class ResetArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("ld", C.c_int32), ("mask", C.c_void_p), ("x", C.c_void_p), ("u", C.c_void_p),
                ("timestep", C.c_void_p), ("weather", C.c_void_p), ("weather_rows", C.c_int32), ("w_off", C.c_void_p),
                ("start_rows", C.c_void_p), ("start_days", C.c_void_p), ("n_starts", C.c_int32),
                ("start_day", C.c_void_p), ("episode", C.c_void_p), ("seed", C.c_uint64)]


class VecNormArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("dim", C.c_int32), ("obs", C.c_void_p), ("obs_out", C.c_void_p),
                ("reward", C.c_void_p), ("reward_out", C.c_void_p), ("done", C.c_void_p), ("obs_mean", C.c_void_p),
                ("obs_var", C.c_void_p), ("obs_count", C.c_void_p), ("ret_stats", C.c_void_p), ("returns", C.c_void_p),
                ("workspace", C.c_void_p), ("gamma", C.c_double), ("epsilon", C.c_double), ("clip_obs", C.c_float),
                ("clip_reward", C.c_float), ("training", C.c_int32), ("norm_obs", C.c_int32),
                ("norm_reward", C.c_int32)]


RULE_FIELDS = ("lamps_on", "lamps_off", "lamps_day_start", "lamps_day_stop", "lamps_off_sun", "lamp_rad_sum_limit",
               "temp_setpoint_day", "temp_setpoint_night", "heat_correction", "heat_deadzone", "co2_day",
               "vent_heat_Pband", "rh_max", "mech_dehumid_Pband", "vent_rh_Pband", "t_vent_off", "vent_cold_Pband",
               "thScrSpDay", "thScrSpNight", "thScrPband", "thScrDeadZone", "thScrRh", "thScrRhPband", "lampExtraHeat",
               "blScrExtraRh", "rhMax", "tHeatBand", "co2Band", "useBlScr")


class RuleCfg(C.Structure):
    _fields_ = [(n, C.c_double) for n in RULE_FIELDS]


class RuleArgs(C.Structure):
    _fields_ = [("B", C.c_int32), ("ld", C.c_int32), ("x", C.c_void_p), ("weather", C.c_void_p),
                ("weather_rows", C.c_int32), ("w_off", C.c_void_p), ("timestep", C.c_void_p), ("start_day", C.c_void_p),
                ("hour", C.c_void_p), ("doy", C.c_void_p), ("control", C.c_void_p)]


class WeatherArgs(C.Structure):
    _fields_ = [("n_raw", C.c_int32), ("time", C.c_void_p), ("i_glob", C.c_void_p), ("t_out", C.c_void_p),
                ("rh", C.c_void_p), ("wind", C.c_void_p), ("t_sky", C.c_void_p), ("co2_ppm", C.c_double),
                ("n_out", C.c_int32), ("nd", C.c_int32), ("out", C.c_void_p), ("workspace", C.c_void_p)]


# every symbol include/glgym.h declares, with its prototype
_DP = C.POINTER(C.c_double)
PROTOTYPES = {
    "glgym_version": (C.c_char_p, []),
    "glgym_abi_version": (C.c_int, []),
    "glgym_set_window": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_layout": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_occupancy": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_ladder_parallel": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_last_error": (C.c_char_p, []),
    "glgym_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _DP, C.c_int, C.c_int, C.c_int,
                               C.POINTER(C.c_void_p)]),
    "glgym_destroy": (C.c_int, [C.c_void_p]),
    "glgym_set_params": (C.c_int, [C.c_void_p, _DP]),
    "glgym_set_params_keep_reward_scale": (C.c_int, [C.c_void_p, _DP]),
    "glgym_set_n_sub": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_model_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_scheme": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_verify": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_reward": (C.c_int, [C.c_void_p, C.POINTER(RewardCfg)]),
    "glgym_get_reward_scale": (C.c_int, [C.c_void_p, _DP, _DP, _DP]),
    "glgym_evalF": (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, C.c_int, C.c_int, _DP]),
    "glgym_rhs": (C.c_int, [C.c_void_p, _DP, _DP, _DP, C.c_int, _DP]),
    "glgym_step": (C.c_int, [C.c_void_p, C.POINTER(StepArgs), C.c_void_p]),
    "glgym_obs": (C.c_int, [C.c_void_p, C.POINTER(ObsArgs), C.c_void_p]),
    "glgym_set_obs_modules": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int]),
    "glgym_obs_dim": (C.c_int, [C.c_void_p, C.c_int]),
    "glgym_set_control_limits": (C.c_int, [C.c_void_p, _DP, _DP, C.c_double]),
    "glgym_reset": (C.c_int, [C.c_void_p, C.POINTER(ResetArgs), C.c_void_p]),
    "glgym_crop_noise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_uint64, C.c_uint64,
                                   C.c_void_p]),
    "glgym_rule_based": (C.c_int, [C.c_void_p, C.POINTER(RuleCfg), C.POINTER(RuleArgs), C.c_void_p]),
    "glgym_vecnorm": (C.c_int, [C.c_void_p, C.POINTER(VecNormArgs), C.c_void_p]),
    "glgym_weather": (C.c_int, [C.c_void_p, C.POINTER(WeatherArgs), C.c_void_p]),
    "glgym_timer_start": (C.c_int, [C.c_void_p, C.c_void_p]),
    "glgym_timer_stop": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]),
}

_lib = None


def load():
    """dlopen libglgym.so and bind every prototype.  Raises GlgymError if the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise GlgymError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        try:
            # PyTorch-ROCm bundles its own HIP runtime; it must be the first one the process loads, otherwise
            # torch.cuda.is_available() turns False once /opt/rocm's copy (our DT_NEEDED) is already resident.
            import torch  # noqa: F401
        except Exception:  # torch is plumbing only; GreenLight.evalF works without it
            pass
        lib = C.CDLL(str(LIB_PATH))
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError here = ABI mismatch with include/glgym.h
            fn.restype, fn.argtypes = res, args
        if lib.glgym_abi_version() != ABI_VERSION:
            raise GlgymError(f"{LIB_PATH} has ABI {lib.glgym_abi_version()}, this binding expects {ABI_VERSION}: rebuild the library")
        _lib = lib
    return _lib


def check(rc: int, what: str = "glgym"):
    if rc != OK:
        msg = load().glgym_last_error().decode() or {EINVAL: "invalid argument", ENODEV: "no HIP device",
                                                       EHIP: "HIP error", ENOMEM: "out of memory"}.get(rc, "")
        raise (GlgymOdeError if rc == EODE else GlgymError)(f"{what} failed (status {rc}): {msg}")
