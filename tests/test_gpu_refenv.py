"""GPU kernels against G3: fixtures produced by the reference's REAL TomatoEnv (tests/golden/make_golden.py g_refenv --
gl_gym/environments/tomato_env.py, observations.py, rewards.py running unmodified over a tight solve of the pinned RHS).

Teacher-forced: env b of one batch replays step b of the fixture episode (state x_b, previous control u_{b-1}, timestep b),
so one launch of step_kernel + obs_kernel covers all 97 steps incl. the terminal one; state errors are then one-step
errors of the sub-stepper against the tight solve and do not compound."""
import numpy as np
import pytest

from conftest import scaled_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("tag", ["rb", "ra"])
def test_step_obs_reward_info_against_reference_env(golden, tag, dtype):
    import torch
    from gl_gym_amd.tomato_env import TomatoVecEnv
    from gl_gym_amd._lib import INFO_KEYS
    g = golden("refenv_1day")
    U, X, OBS, R, INFO, DONE = (g[f"{tag}_{k}"] for k in ("u", "x", "obs", "reward", "info", "done"))
    B = len(U)
    assert B == 97 and list(g["info_keys"]) == list(INFO_KEYS)
    env = TomatoVecEnv(B, weather=g["weather"], params=g["p"], dtype=dtype, season_length=1, pred_horizon=0.5,
                       start_rows=[0], start_days=[0.0], auto_reset=False)
    obs0 = env.reset()
    np.testing.assert_allclose(obs0[0], OBS[0], rtol=2e-6, atol=2e-6)            # reset observation (f32 block)
    assert env.N == int(g["N"]) and env.Np == int(g["Np"]) and env.obs_dim == 263
    dev, T = env.device, env.tdtype
    env.x.copy_(torch.as_tensor(X[:B], dtype=T, device=dev))
    u_prev = np.vstack([np.zeros((1, 6)), U[:-1]])
    env.u.copy_(torch.as_tensor(u_prev, dtype=T, device=dev))
    env.timestep_t.copy_(torch.arange(B, dtype=torch.int32, device=dev))
    if tag == "rb":
        obs, r, done, infos = env.step_raw_control(U)
    else:
        obs, r, done, infos = env.step(g["ra_actions"][:B])
        np.testing.assert_allclose(env.u.double().cpu().numpy(), U, rtol=0, atol=1e-7 if dtype == "float32" else 1e-15)
    x1 = env.x.double().cpu().numpy()
    e_x = scaled_err(x1, X[1:B + 1])
    assert e_x < 1e-4, e_x                                                        # one step vs the tight solve
    # observations: 4 + 3 state-derived entries carry the sub-stepper's error; controls, weather, clocks and the
    # 240-entry forecast block are exact up to float32 rounding (observations.py:59-182)
    ref = OBS[1:B + 1]
    np.testing.assert_allclose(obs[:, 7:], ref[:, 7:], rtol=3e-6, atol=3e-6)
    sc = np.maximum(np.abs(ref[:, :7]), 1e-3 * np.abs(ref[:, :7]).max(axis=0))
    assert np.max(np.abs(obs[:, :7] - ref[:, :7]) / sc) < 2e-4
    np.testing.assert_array_equal(done, DONE)
    assert done[-1] and not done[:-1].any()                                       # episode = N + 1 steps
    # reward / info: the fruit-gain term carries the one-step error of cFruit (5e4 mg m-2 scale, times 2.5e-5)
    assert np.max(np.abs(r - R)) < 2e-4
    info_gpu = (np.asarray(infos)[:, :B].T if isinstance(infos, np.ndarray)            # step_raw_control: [11, B] block
                else np.array([[infos[b][q] for q in INFO_KEYS] for b in range(B)]))
    np.testing.assert_allclose(info_gpu[:, 2:7], INFO[:, 2:7], rtol=2e-6, atol=1e-9)   # costs: functions of u only
    np.testing.assert_allclose(info_gpu[:, 0:2], INFO[:, 0:2], rtol=0, atol=3e-6)      # EPI, revenue
    viol_sc = np.array([15.0, 2500.0, 15.0, 1.0])
    assert np.max(np.abs(info_gpu[:, 7:11] - INFO[:, 7:11]) / viol_sc) < 2e-4
    print(f"refenv {tag} {dtype}: state {e_x:.2e}, reward {np.max(np.abs(r - R)):.2e}")
    env.close()
