# Quick PMC look at the default fp32 RK4 step kernel on the bench workload (sustained phase: 60 warm-up steps, 40 profiled steps).
export TMPDIR=/tmp
OUT=gpurun_out/r04q; mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-alt-scheme --no-parity "$@" > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT/pmc*/*/*counter_collection.csv > $OUT/pmc_summary.csv
grep step_kernel $OUT/pmc_summary.csv
rm -rf $OUT/pmc?/
