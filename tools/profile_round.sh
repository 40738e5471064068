export TMPDIR=/tmp
set -x
mkdir -p gpurun_out/v8
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/v8/stats -- python3 bench.py --steps 20 --warmup 3 > gpurun_out/v8/bench_stats.log 2>&1
grep "^{" gpurun_out/v8/bench_stats.log > gpurun_out/v8/bench_line.json
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/v8/pmc$i -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/v8/pmc$i.log 2>&1 || echo "pass $i failed"
done
f=$(ls gpurun_out/v8/stats/*/*kernel_stats.csv | head -1); python tools/condense_stats.py $f gpurun_out/v8/kernel_stats.csv
python tools/pmc_summary.py gpurun_out/v8/pmc*/*/*counter_collection.csv > gpurun_out/v8/pmc_summary.csv
cat gpurun_out/v8/kernel_stats.csv | head -8
grep step_kernel gpurun_out/v8/pmc_summary.csv
