#!/bin/bash
# GPU job B of round 6: the fp64 four-lanes-per-environment kernel before / after the quad shares its transcendental slots
# (gl_model_quad.hpp gq_stage, SHARE_POW): timings (config 2, B = 8, per-stage / per-window fits, evalF latency) and the fp64 parity tests.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PKG=$PWD/greenlight-gym2_amd/gl_gym_amd
O=gpurun_out/r06_job_b.txt; : > $O
for lib in libglgym_pre_l1.so libglgym.so; do
  echo "==== $lib" >> $O
  export GLGYM_LIB=$PKG/$lib
  for B in 8 4096; do python tools/window_cost.py $B float64 >> $O 2>&1; done
  python tools/window_cost.py 8 float32 quad 2>&1 | grep -v "^rk" | head -8 >> $O
  for args in "--dtype f64 --batch 4096" "--dtype f64 --batch 8" "--dtype f64 --batch 65536 --steps 60 --warmup 10" "--batch 8" "--batch 4096" "--batch 16384"; do
    python bench.py $args --steps ${STEPS:-200} --warmup 20 --no-cpu-baseline --no-alt-scheme --no-parity --no-parity-config --no-sustained 2>/dev/null | grep "^{" | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench $args: %.4g env-steps/s, %.4f ms/step, kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> $O
  done
  python tools/evalf_latency.py 100 2>&1 | grep -E "float64 +parity|float32 +throughput|GreenLight.evalF" | grep -E " (1|8) +[0-9]|GreenLight" >> $O
done
unset GLGYM_LIB
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_storm.py tests/test_gpu_fuzz.py tests/test_gpu_jump.py tests/test_gpu_refenv.py tests/test_gpu_env_api.py -q -m gpu -k "not test_10day and not config3 and not bench_workload and not full_size and not two_waves" > gpurun_out/r06_gputest_b.log 2>&1
timeout 900 python -m pytest tests/test_gpu_holdout.py -q -m gpu -k "f64" >> gpurun_out/r06_gputest_b.log 2>&1
tail -4 gpurun_out/r06_gputest_b.log
cat $O
