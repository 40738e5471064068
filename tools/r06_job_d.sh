#!/bin/bash
# GPU job D of round 6: harvest_flow with wave-uniform early exits against the build before it: bit comparison, bench lines, per-window fit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PKG=$PWD/greenlight-gym2_amd/gl_gym_amd
GLGYM_LIB=$PKG/libglgym_pre_hv.so timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_before.npz > gpurun_out/r06_bitcompare_d.log 2>&1
timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare_d.log 2>&1
python tools/lib_bitcompare.py compare gpurun_out/bits_before.npz gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare_d.log 2>&1
rm -f gpurun_out/bits_before.npz gpurun_out/bits_after.npz
O=gpurun_out/r06_job_d.txt; : > $O
for rep in 1 2; do for lib in libglgym_pre_hv.so libglgym.so; do
  export GLGYM_LIB=$PKG/$lib
  for args in "" "--uncertainty 0.2" "--batch 262144 --steps 300 --warmup 30" "--batch 8" "--dtype f64 --batch 4096 --steps 200"; do
    python bench.py $args --no-cpu-baseline --no-alt-scheme --no-parity --no-parity-config --no-sustained 2>/dev/null | grep "^{" | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib bench $args: %.4g env-steps/s, %.4f ms/step, kernel %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" >> $O
  done
done; done
for lib in libglgym_pre_hv.so libglgym.so; do export GLGYM_LIB=$PKG/$lib; echo "== $lib" >> $O; python tools/window_cost.py 65536 float32 one 2>&1 | grep -E "^ls5|fit" | head -6 >> $O; done
unset GLGYM_LIB
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "crop or harvest or config5 or step_kernel_matches or evalF_signature" > gpurun_out/r06_gputest_d.log 2>&1
tail -3 gpurun_out/r06_bitcompare_d.log; tail -3 gpurun_out/r06_gputest_d.log; cat $O
