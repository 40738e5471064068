export TMPDIR=/tmp
mkdir -p gpurun_out/r02c
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-scheme > gpurun_out/r02c/calm.log 2>&1
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/r02c/pmc$i -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-scheme > gpurun_out/r02c/pmc$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py gpurun_out/r02c/pmc*/*/*counter_collection.csv > gpurun_out/r02c/pmc_summary.csv
grep step_kernel gpurun_out/r02c/pmc_summary.csv
grep "^{" gpurun_out/r02c/calm.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('calm', d['value'], d['roofline']['kernel_ms'])"
rm -rf gpurun_out/r02c/pmc?/
