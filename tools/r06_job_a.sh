#!/bin/bash
# GPU job A of round 6: (1) the hold-out fixtures on the AS-SHIPPED round-5 library (libglgym_r05.so = the binary of commit ecf928e),
# (2) the same on the current build, (3) bit-comparison of the build before / after the sc_policy.hpp refactor, (4) the GPU suite,
# (5) bench lines (driver-style short run, default run).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
PKG=$PWD/greenlight-gym2_amd/gl_gym_amd
rm -f gpurun_out/r06_holdout.txt
GLGYM_LIB=$PKG/libglgym_r05.so timeout 1200 python -m pytest tests/test_gpu_holdout.py -q -m gpu > gpurun_out/r06_holdout_as_shipped_pytest.log 2>&1
mv gpurun_out/r06_holdout.txt gpurun_out/r06_holdout_as_shipped.txt
timeout 1200 python -m pytest tests/test_gpu_holdout.py -q -m gpu > gpurun_out/r06_holdout_pytest.log 2>&1
GLGYM_LIB=$PKG/libglgym_q4.so timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_before.npz > gpurun_out/r06_bitcompare.log 2>&1
timeout 600 python tools/lib_bitcompare.py dump gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare.log 2>&1
python tools/lib_bitcompare.py compare gpurun_out/bits_before.npz gpurun_out/bits_after.npz >> gpurun_out/r06_bitcompare.log 2>&1
rm -f gpurun_out/bits_before.npz gpurun_out/bits_after.npz
timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_gpu_holdout.py -s > gpurun_out/r06_gputest_a.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_style.json 2> gpurun_out/r06_bench_driver_style.err
timeout 300 python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
tail -3 gpurun_out/r06_holdout_as_shipped_pytest.log gpurun_out/r06_holdout_pytest.log gpurun_out/r06_bitcompare.log gpurun_out/r06_gputest_a.log
