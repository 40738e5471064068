set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python bench.py > gpurun_out/r05/bench_full.log 2>&1; grep "^{" gpurun_out/r05/bench_full.log > gpurun_out/r05/r05_bench_line_default.json
python bench.py --steps 12000 --warmup 500 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_soak_ls5_bench_line.json
python tools/substep_hist.py 300 > gpurun_out/r05/r05_substep_hist.txt 2>&1
python tools/evalf_latency.py 300 > gpurun_out/r05/r05_evalf_latency.txt 2>&1
bash tools/bench_variants_r05.sh > gpurun_out/r05/variants.log 2>&1
cat gpurun_out/r05/variants.log
for B in 131072 262144 524288 1048576; do for occ in 1 2; do GLGYM_OCC=$occ python bench.py --batch $B --steps 100 --warmup 30 --no-cpu-baseline --no-alt-scheme --no-parity-config --no-parity 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"B\", $B, \"occ\", $occ, \"%.4g env-steps/s\" % d[\"value\"], \"kernel %.3f ms\" % d[\"roofline\"][\"kernel_ms\"])"; done; done
python bench.py --batch 262144 --steps 3000 --warmup 100 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_soak_b262144_bench_line.json
for s in rk4 rk3 rk2; do python bench.py --scheme $s --steps 12000 --warmup 500 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_soak_${s}_bench_line.json; done
python -m pytest tests -m gpu -q -s > gpurun_out/r05/gputest_full.log 2>&1; tail -1 gpurun_out/r05/gputest_full.log
