"""Hold-out fixtures (round 6) on the CPU: what they are, and the PRODUCT's gl_model.hpp arithmetic (host instantiation,
tests/hostmath -- fp64, and the generic template in float) free-running on them with the constants as shipped.  The GPU kernels
are checked through the C ABI in tests/test_gpu_holdout.py, whose docstring has the bar and the frost finding."""
import numpy as np
import pytest

from conftest import judge_rollout as judge


def test_fixtures_are_what_they_say(golden):
    p0 = golden("params_default")["p"]
    g = golden("holdout_gl2010_random")
    assert g["X"].shape == (962, 28) and g["actions"].shape == (961, 6) and g["actions"].dtype == np.float32
    assert g["weather"][:961, 1].min() < -8 and g["weather"][:961, 1].max() < 6.5           # the frost fortnight of GL2010, not GL2009's autumn
    assert not np.allclose(g["weather"][:100], golden("rollout_10day")["weather"][:100])
    u = np.zeros(6)
    for k in range(961):                                                                     # U is what tomato_env.py:113 makes of the actions
        u = np.clip(u + g["actions"][k] * np.float32(0.1), np.float32(0), np.float32(1))
        assert np.array_equal(u, g["U"][k])
    g = golden("holdout_gl2010_rulebased")
    assert g["U"].shape == (961, 6) and np.array_equal(g["p"], p0)
    assert (np.abs(np.diff(g["U"], axis=0)).max(axis=1) > 0.5).sum() > 400                   # bang-bang: raw control jumps in most steps
    assert (g["U"][:, 4] > 0.5).sum() > 500                                                  # lamps on most of the time (winter)
    g = golden("holdout_runtime_dt300")
    p = g["p"]
    assert g["U"].shape == (2881, 6) and g["X"].shape == (961, 28)
    changed = np.nonzero(p != p0)[0]
    assert set(changed.tolist()) == {79, 108, 145, 165, 170} and p[109] == 720.0             # gl_predefined_controls.py:70-77 (its p[109] = 720 IS the default)
    assert p[145] == 300000 and list(g["x0"][22:27]) == [0.0, 0.9e5, 2.5e5, 2.8e5, 3000.0]   # run_time.py:44 set_crop_state
    g = golden("holdout_season60")
    assert g["X"].shape == (8, 62, 28) and g["actions_q"].shape == (8, 5761, 6) and g["actions_q"].dtype == np.int8
    assert list(g["kept_steps"][:3]) == [0, 96, 192] and g["kept_steps"][-1] == 5761
    assert np.all(np.isfinite(g["X"])) and np.all(g["X"][:, -1, 26] > g["X"][:, 0, 26] + 500)       # 60 days of canopy temperature summed
    assert len({tuple(g["X"][b, -1, 22:26].round(0)) for b in range(8)}) == 8                       # eight DIFFERENT seasons
    g = golden("holdout_gl2010_noisy")
    assert g["X"].shape == (4, 962, 28) and g["P_crop"].shape == (4, 961, 34) and g["P_crop"].dtype == np.float32 and np.array_equal(g["p"], p0)
    rel = np.abs(g["P_crop"] / p0[128:162] - 1)
    assert np.all(rel[:, :, np.arange(34) != 16] <= 0.1 + 1e-6) and rel.max() > 0.09             # noise.py: +-10 % at scale 0.2; p[144] is derived
    assert np.allclose(g["P_crop"][..., 16], g["P_crop"][..., 13] / g["P_crop"][..., 14], rtol=1e-6)           # cLeafMax = laiMax / sla
    assert len({g["P_crop"][b, k].tobytes() for b in range(4) for k in range(961)}) == 4 * 961                 # a new block at EVERY step
    assert not np.allclose(g["weather"][:100], golden("holdout_gl2010_random")["weather"][:100])               # day 40, not day 20


@pytest.mark.parametrize("name,dt,verify,stride", [("holdout_gl2010_random", 900.0, False, 1), ("holdout_gl2010_rulebased", 900.0, True, 1),
                                                   ("holdout_runtime_dt300", 300.0, True, 3)])
def test_product_arithmetic_on_the_holdouts_with_the_constants_as_shipped(golden, hostmath, name, dt, verify, stride):
    """fp64 host instantiation of the shipped sub-stepper (ls5; throughput 128 / window 2 and parity 192 / window 1 at dt = 900 s,
    scaled with dt), free-running over the whole fixture: nothing above the bar away from the freezing point, none failed; the parity
    preset inside the band a BDF solve at the reference's tolerances keeps from the same truth.  Plain metric on the run_time fixture at
    the throughput preset: 1.07e-4 -- one step of 2 881, tTop = -0.0048 C off by 1.8e-6 K, where that BDF solve is at 2.6e-4."""
    g = golden(name)
    w, XR, U = g["weather"], g["X"], g["U"].astype(np.float64)
    p = (g["p"] if "p" in g.files else golden("params_default")["p"]).astype(np.float64)
    x0 = g["x0"] if "x0" in g.files else XR[0]
    band = float(g["bdf_free"].max())
    for preset, (n0, win) in (("throughput", (128, 2)), ("parity", (192, 1))):
        n_sub = max(win, int(-(-(n0 * dt / 900.0) // win) * win))
        x, failed, X = x0.copy(), 0, [x0.copy()]
        for k in range(len(U)):
            x, retries, extra, bad = hostmath.step_guarded(x, U[k], w[k], p, dt=dt, n_sub=n_sub, order=5, window=win, verify=verify)
            failed += bad
            if (k + 1) % stride == 0:
                X.append(x.copy())
        plain, who, step, real, floor = judge(np.array(X), XR[:len(X)], abs_floor=1e-4)
        print(f"{name} host fp64 ls5 {preset} n_sub {n_sub}: plain metric {plain:.2e} ({who}, kept step {step}), above the bar away from 0 C: {real}, "
              f"at the floor: {floor}; failed {failed}; BDF-1e-6 band {band:.2e}")
        assert failed == 0 and real == 0
        assert plain < (2e-5 if preset == "parity" else max(1.1e-4, 0.5 * band))
        if preset == "parity":
            assert plain < band


def test_product_arithmetic_on_the_noisy_parameter_holdout(golden, hostmath):
    """The same for the per-step crop-parameter fixture (environment 0): the host fp64 instantiation is handed the block each step was solved with."""
    g = golden("holdout_gl2010_noisy")
    w, XR, U, P = g["weather"], g["X"][0], g["U"][0].astype(np.float64), g["P_crop"][0].astype(np.float64)
    band = float(g["bdf_free"][0].max())
    p = g["p"].astype(np.float64)
    for preset, (n_sub, win) in (("throughput", (128, 2)), ("parity", (192, 1))):
        x, failed, X = XR[0].copy(), 0, [XR[0].copy()]
        for k in range(len(U)):
            p[128:162] = P[k]
            x, retries, extra, bad = hostmath.step_guarded(x, U[k], w[k], p, dt=900.0, n_sub=n_sub, order=5, window=win, verify=False)
            failed += bad
            X.append(x.copy())
        plain, who, step, real, floor = judge(np.array(X), XR, abs_floor=1e-4)
        print(f"holdout_gl2010_noisy env 0 host fp64 ls5 {preset}: plain metric {plain:.2e} ({who}, step {step}), above the bar away from 0 C: {real}, "
              f"at the floor: {floor}; failed {failed}; BDF-1e-6 band {band:.2e}")
        assert failed == 0 and real == 0
        assert plain < (3e-5 if preset == "parity" else 2e-4)       # throughput: 1.6e-4 on THREE steps (grow-pipe temperature -0.05 C, off by 8e-6 K; BDF-1e-6: 1.4e-4); parity 2.1e-5
        if preset == "parity":
            assert plain < band


REF_WEATHER = "/root/reference/gl_gym/environments/weather"


@pytest.mark.skipif(not __import__("os").path.isdir(REF_WEATHER), reason="the reference's weather files are only in the build container")
def test_host_weather_loader_follows_the_reference_past_the_end_of_a_file(golden):
    """refenv_day60.npz holds the table the REFERENCE's loader produced for start day 60 of GL2009 with the 49 days of horizon its env asks
    for (tomato_env.py:250-260): it runs past the end of the file and appends GL2010 (utils.py expandWeatherData).  The package's host
    loader reads the same two files (data, not code) and must reproduce the table bit for bit."""
    from gl_gym_amd.utils import load_weather_data
    g = golden("refenv_day60")
    w = load_weather_data(REF_WEATHER, "Bleiswijk", "GL", 2009, 60, 2, int(g["Np"]) + 1, 900, 10)
    assert w.shape == g["weather"].shape == (4896, 10)
    np.testing.assert_array_equal(w, g["weather"])
