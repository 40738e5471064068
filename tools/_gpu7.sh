set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q -x 2>&1 | tail -15
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['parity']['max_scaled_err_10day'], d['integrator_events'], 'rk4', d['other_scheme']['value'], 'pc', d['parity_config']['value'], d['parity_config']['max_scaled_err_10day'])"
python tools/substep_hist.py 300 2>&1 | grep -v Warn | tail -7
python bench.py --dtype f64 --batch 4096 --steps 300 --warmup 50 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'], d['parity']['max_scaled_err_10day'])"
