set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_var
for B in 131072 262144 524288; do python bench.py --batch $B --steps 300 --warmup 30 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05_var/r05_variant_b${B}_bench_line.json; python -c "
import json; d=json.load(open('gpurun_out/r05_var/r05_variant_b${B}_bench_line.json')); r=d['roofline']; print('B', $B, '%.4g'%d['value'], 'kernel %.3f'%r['kernel_ms'], 'frac', r['frac'], r.get('frac_packed_weighted'), r.get('pmc_variant'), 'failed', d['integrator_events']['failed_integrations'])"; done
python bench.py --steps 1000 --warmup 100 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default', d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
