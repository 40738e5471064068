set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
ONLY='^f32_ls5_occ2$' bash tools/profile_r05.sh > gpurun_out/r05/profile_occ2.log 2>&1; tail -3 gpurun_out/r05/profile_occ2.log
cp gpurun_out/r05_prof/f32_ls5_occ2/constants.json gpurun_out/r05/occ2_constants.json
for B in 131072 262144 524288; do python bench.py --batch $B --steps 200 --warmup 30 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_variant_b${B}_bench_line.json; python -c "
import json; d=json.load(open('gpurun_out/r05/r05_variant_b${B}_bench_line.json')); print('B', $B, '%.4g'%d['value'], 'kernel %.3f'%d['roofline']['kernel_ms'], 'parity', d['parity']['max_scaled_err_10day'], 'events', d['integrator_events']['first_attempt_flags'], d['integrator_events']['failed_integrations'])"; done
