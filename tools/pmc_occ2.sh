#!/bin/bash
# Diagnostic (round 5): why do two co-resident waves of the default kernel run at 0.89x of two sequential ones?  PMC passes of the
# one-wave and the two-wave build at B = 131 072 (two waves per SIMD either way: sequential / co-resident).
export TMPDIR=/tmp
OUT=gpurun_out/r05_occ2; mkdir -p $OUT
for occ in ${OCCS:-1 2}; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    GLGYM_OCC=$occ timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/occ${occ}_p$i -- python3 bench.py --batch 131072 --steps 20 --warmup 30 --no-cpu-baseline --no-alt-scheme --no-parity --no-parity-config > $OUT/occ${occ}_p$i.log 2>&1 || echo "occ $occ pass $i failed"
  done
  python tools/pmc_summary.py $OUT/occ${occ}_p*/*/*counter_collection.csv | grep "step_kernel" > $OUT/r05_occ${occ}_b131072_pmc_summary.csv
  rm -rf $OUT/occ${occ}_p?
done
cat $OUT/r05_occ1_b131072_pmc_summary.csv $OUT/r05_occ2_b131072_pmc_summary.csv
