set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench_full.log 2>&1; tail -c 6000 gpurun_out/r05/bench_full.log | grep "^{" > gpurun_out/r05/r05_bench_line_default.json
python tools/evalf_latency.py 300 > gpurun_out/r05/r05_evalf_latency.txt 2>&1
ONLY='^(f32_ls5|f64_ls5_quad)$' bash tools/profile_r05.sh > gpurun_out/r05/profile.log 2>&1
tail -5 gpurun_out/r05/profile.log
cat gpurun_out/r05/r05_evalf_latency.txt
cat gpurun_out/r05/r05_bench_line_default.json
