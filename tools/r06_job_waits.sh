#!/bin/bash
# What the lone wavefront of the four-lanes-per-environment kernels waits for (B = 8): LDS / scalar-memory / vector-memory instruction counts and
# in-flight levels (average latency = LEVEL / INSTS), tools/window_one.py under rocprofv3 --pmc.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r06_waits.txt; rm -f $out
for dt in float64 float32; do
  for set in "SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_WAIT_INST_ANY"; do
    rm -rf gpurun_out/pmcic
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmcic -- python3 tools/window_one.py $dt 2 > /dev/null 2>&1
    python tools/pmc_summary.py gpurun_out/pmcic/*/*counter_collection.csv 2>/dev/null | grep -i "step_kernel" | sed "s/^/$dt window 2: /" >> $out
  done
done
rm -rf gpurun_out/pmcic
cat $out
