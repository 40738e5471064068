#!/bin/bash
# GPU job C of round 6: the build with the two-rungs-at-a-time STEP kernels (verified raw-control steps at small batches):
# bit-comparison against the previous build, the whole GPU suite, raw-control step latency with the ladder sequential / paired, and
# the vector instructions of one tier-2b window from the PMC (same stages, window 1 vs 2).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PKG=$PWD/greenlight-gym2_amd/gl_gym_amd
GLGYM_LIB=$PKG/libglgym_pre_l1.so timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_before.npz > gpurun_out/r06_bitcompare_c.log 2>&1
timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare_c.log 2>&1
python tools/lib_bitcompare.py compare gpurun_out/bits_before.npz gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare_c.log 2>&1
rm -f gpurun_out/bits_before.npz gpurun_out/bits_after.npz
rm -f gpurun_out/r06_holdout.txt
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r06_gputest_c.log 2>&1
python - > gpurun_out/r06_raw_control_latency.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, "greenlight-gym2_amd")
import numpy as np, torch
from gl_gym_amd.tomato_env import TomatoVecEnv
from gl_gym_amd.utils import synthetic_weather
w = synthetic_weather(n_rows=4000)
print("# glgym_step(control = ...) -- step_raw_control, verified -- per env-step, device tensors in and out (median of 200 after 30); MI355X")
print("# dtype   preset      B     sequential ladder [ms]   two rungs at a time [ms]")
for dtype in ("float64", "float32"):
    for preset in ("parity", "throughput"):
        for B in (1, 8, 64, 1024, 8192):
            row = []
            for par in (False, True):
                env = TomatoVecEnv(B, weather=w, dtype=dtype, preset=preset, season_length=10, start_rows=[0, 96, 480], seed=3, auto_reset=False)
                env.set_ladder_parallel(par)
                env.reset_tensor()
                g = torch.Generator(device=env.device).manual_seed(1)
                ctrl = torch.rand(B, 6, generator=g, device=env.device, dtype=env.tdtype)
                ts = []
                for i in range(230):
                    ctrl = (ctrl + 0.05 * (torch.rand(B, 6, generator=g, device=env.device, dtype=env.tdtype) - 0.5)).clamp_(0, 1)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    env.step_tensor(controls_t=ctrl, want_obs=False)
                    torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                row.append(1e3 * float(np.median(ts[30:])))
                env.close()
            print(f"{dtype:8s} {preset:10s} {B:5d}   {row[0]:10.3f}   {row[1]:10.3f}   ({row[0] / row[1]:.2f}x)", flush=True)
PY
export TMPDIR=/tmp
for dt in float64 float32; do for win in 1 2; do
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d gpurun_out/pmcwin_${dt}_$win -- python3 tools/window_one.py $dt $win > /dev/null 2>&1
  python tools/pmc_summary.py gpurun_out/pmcwin_${dt}_$win/*/*counter_collection.csv 2>/dev/null | grep -i "step_kernel" | head -3 | sed "s/^/$dt window $win: /" >> gpurun_out/r06_window_insts.txt
  rm -rf gpurun_out/pmcwin_${dt}_$win
done; done
tail -3 gpurun_out/r06_bitcompare_c.log; tail -5 gpurun_out/r06_gputest_c.log; cat gpurun_out/r06_raw_control_latency.txt; cat gpurun_out/r06_window_insts.txt
