#!/bin/bash
# Does aligning branch targets change the lone-wave kernels?  (SQ_WAIT_INST_ANY: 6-12 % of the wave cycles in the two-rungs evalF kernels.)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
out=gpurun_out/r06_align_ab.txt; rm -f $out
for lib in greenlight-gym2_amd/gl_gym_amd/libglgym.so "$@"; do
  [ -f $lib ] || continue
  echo "== $lib" >> $out
  GLGYM_LIB=$PWD/$lib timeout 200 python tools/evalf_ab_fp64.py 2>&1 | grep -v amdgpu.ids | grep "pair auto 1\|seq  auto 1\|never 1" >> $out
  for cfg in "--batch 65536" "--batch 4096 --dtype f64" "--batch 8 --dtype f64" "--batch 8"; do
    GLGYM_LIB=$PWD/$lib timeout 300 python bench.py --steps 600 --warmup 100 --no-sustained --no-cpu-baseline --no-parity --no-parity-config --no-alt-scheme $cfg 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print('bench $cfg: value %.4e ms_per_step %.4f kernel_ms %s' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms')))
" >> $out
  done
done
cat $out
