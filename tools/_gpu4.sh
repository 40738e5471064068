set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q -s 2>&1 | tee gpurun_out/r05/gputest_full.log | grep -E "passed|failed|error|^(fuzz|jump|storm|stress|10-day|evalF|config)" | tail -80
python bench.py > gpurun_out/r05/bench_full.log 2>&1; grep "^{" gpurun_out/r05/bench_full.log > gpurun_out/r05/r05_bench_line_default.json
python bench.py --steps 12000 --warmup 500 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_soak_ls5_bench_line.json
bash tools/bench_variants_r05.sh > gpurun_out/r05/variants.log 2>&1
python tools/substep_hist.py 300 > gpurun_out/r05/r05_substep_hist.txt 2>&1
python tools/evalf_latency.py 300 > gpurun_out/r05/r05_evalf_latency.txt 2>&1
python tools/small_batch_rate.py float32 > gpurun_out/r05/r05_small_batch_rate_fp32.txt 2>&1
python tools/small_batch_rate.py float64 > gpurun_out/r05/r05_small_batch_rate_fp64.txt 2>&1
bash tools/profile_r05.sh > gpurun_out/r05/profile.log 2>&1
tail -12 gpurun_out/r05/profile.log
cat gpurun_out/r05/variants.log
