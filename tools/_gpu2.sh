set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -x -q -k "fp64 or float64 or f64 or fuzz or rhs or kat" 2>&1 | tail -5
for i in 1 2 3; do python bench.py --dtype f64 --batch 4096 --steps 300 --warmup 50 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'], d['parity']['max_scaled_err_10day'], d['roofline'].get('frac'))"; done
python bench.py --dtype f64 --steps 100 --warmup 20 --no-cpu-baseline --no-alt-scheme --no-parity-config --no-parity 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('f64 B65536', d['value'], d['ms_per_step'])"
GLGYM_OCC=2 python bench.py --batch 262144 --steps 200 --warmup 50 --no-cpu-baseline --no-alt-scheme --no-parity-config --no-parity 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ2 B262144', d['value'], d['ms_per_step'])"
python bench.py --batch 262144 --steps 200 --warmup 50 --no-cpu-baseline --no-alt-scheme --no-parity-config --no-parity 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ1 B262144', d['value'], d['ms_per_step'], d['roofline'].get('frac'))"
