#!/bin/bash
# GPU job E of round 6: after the evalF fp32 regression fix (ladder bookkeeping compiled out of glgym_evalF, the pre-round-6 harvest_flow
# body for the fp32 kernels with SGPR-resident parameters): evalF latency against the round-5 binary on the same box, the GPU suite.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PKG=$PWD/greenlight-gym2_amd/gl_gym_amd
GLGYM_LIB=$PKG/libglgym_r05.so python tools/evalf_ab.py > gpurun_out/r06_evalf_ab.txt 2>&1
python tools/evalf_ab.py >> gpurun_out/r06_evalf_ab.txt 2>&1
rm -f gpurun_out/r06_holdout.txt
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r06_gputest_e.log 2>&1
python bench.py --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench default: %.4g env-steps/s, kernel %.4f ms, frac %.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['frac']))" > gpurun_out/r06_bench_e.txt
grep -v amdgpu gpurun_out/r06_evalf_ab.txt | grep -E "float32 throughput|float64 parity pair auto"; tail -2 gpurun_out/r06_gputest_e.log; cat gpurun_out/r06_bench_e.txt
