#!/bin/bash
# Instruction-cache behaviour of the two-rungs-at-a-time evalF kernels (B = 1, verified): tools/evalf_one.py under rocprofv3 --pmc.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_icache_evalf.txt; rm -f $out
for lib in libglgym.so libglgym_r05.so; do for cfg in "float64 parity" "float64 parity seq" "float32 parity"; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    rm -rf gpurun_out/pmcic
    GLGYM_LIB=$PWD/greenlight-gym2_amd/gl_gym_amd/$lib rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcic -- python3 tools/evalf_one.py $cfg > /dev/null 2>&1
    python tools/pmc_summary.py gpurun_out/pmcic/*/*counter_collection.csv 2>/dev/null | grep -i "evalf_kernel" | sed "s/^/$lib $cfg: /" >> $out
  done
done; done
rm -rf gpurun_out/pmcic
cat $out
