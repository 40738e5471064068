p() { grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '%.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'kernel_ms %.3f' % d['roofline']['kernel_ms'], 'fail', d['ode_failures'])"; }
python bench.py --no-cpu-baseline 2>&1 | p default_sustained
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --uncertainty 0.2 2>&1 | p config5
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --vecnorm 2>&1 | p vecnorm
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --dtype f64 --batch 4096 2>&1 | p f64_4096
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --dtype f64 --batch 65536 2>&1 | p f64_65536
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --batch 524288 2>&1 | p B524288
for n in 224 320 512; do python bench.py --steps 10 --warmup 2 --no-cpu-baseline --n-sub $n 2>&1 | p nsub$n; done
python tools/host_path_rate.py 2>&1 | tail -6
