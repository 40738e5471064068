#!/bin/bash
# Final GPU jobs of round 6 (2/2): bench lines (default, driver-style short run, soak), variants, small-batch and latency tables, the
# GPU test printout -- on the final code, after profiles/r06_pmc_constants.json has been updated.
set -x
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r06
python bench.py > gpurun_out/r06/bench_full.log 2>&1; grep "^{" gpurun_out/r06/bench_full.log > gpurun_out/r06/r06_bench_line_default.json
python bench.py --steps 20 --warmup 5 2>/dev/null | grep "^{" > gpurun_out/r06/r06_bench_line_driver_style.json
python bench.py --steps 12000 --warmup 500 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r06/r06_soak_ls5_bench_line.json
python bench.py --scheme rk4 --steps 12000 --warmup 500 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r06/r06_soak_rk4_bench_line.json
python bench.py --batch 262144 --steps 3000 --warmup 100 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r06/r06_soak_b262144_bench_line.json
python tools/substep_hist.py 300 > gpurun_out/r06/r06_substep_hist.txt 2>&1
python tools/evalf_latency.py 300 > gpurun_out/r06/r06_evalf_latency.txt 2>&1
bash tools/bench_variants_r06.sh > gpurun_out/r06/variants.log 2>&1
cat gpurun_out/r06/variants.log
python tools/small_batch_rate.py float32 > gpurun_out/r06/r06_small_batch_rate_fp32.txt 2>&1
GLGYM_RATE_SIZES=8,64,1024,4096 python tools/small_batch_rate.py float64 > gpurun_out/r06/r06_small_batch_rate_fp64.txt 2>&1
python -m pytest tests -m gpu -q -s > gpurun_out/r06/gputest_full.log 2>&1; tail -1 gpurun_out/r06/gputest_full.log
