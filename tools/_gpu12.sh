set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()"
python bench.py > gpurun_out/r05/bench_final.log 2>&1; grep "^{" gpurun_out/r05/bench_final.log > gpurun_out/r05/r05_bench_line_default.json
python -c "
import json; d=json.load(open('gpurun_out/r05/r05_bench_line_default.json')); r=d['roofline']; print('final', d['value'], d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_packed_weighted'), d['parity']['max_scaled_err_10day'], d['parity_config']['value'], d['other_scheme']['value'], d['cpu_baseline']['value'])"
