#!/bin/bash
# Instruction-cache behaviour of the four-lanes-per-environment kernels at B = 8 (one wavefront): tools/window_one.py under rocprofv3 --pmc.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_icache.txt; rm -f $out
for dt in float64 float32; do for win in 1 2; do
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"; do
    rm -rf gpurun_out/pmcic
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcic -- python3 tools/window_one.py $dt $win > /dev/null 2>&1
    python tools/pmc_summary.py gpurun_out/pmcic/*/*counter_collection.csv 2>/dev/null | grep -i "step_kernel" | sed "s/^/$dt window $win: /" >> $out
  done
done; done
rm -rf gpurun_out/pmcic
cat $out
