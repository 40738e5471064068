#!/bin/bash
# Round 6: bench lines of the other configurations (BASELINE configs[1], [4], large and small batches, the other schemes), default step
# counts shortened.  Every line is a full bench.py line (roofline, parity, parity_config); the CPU-baseline legs are skipped.
OUT=gpurun_out/r06_var; mkdir -p $OUT
run() { name=$1; shift; python3 bench.py --no-cpu-baseline --no-alt-scheme "$@" 2> $OUT/$name.err | grep "^{" > $OUT/r06_variant_${name}_bench_line.json; python3 -c "
import json,sys; d=json.load(open('$OUT/r06_variant_${name}_bench_line.json')); pc = d.get('parity_config') or {}
print('$name', '%.4g env-steps/s' % d['value'], '%.4f ms/step' % d['ms_per_step'], 'kernel %.4f ms' % d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'parity', (d.get('parity') or {}).get('max_scaled_err_10day'), '| parity config', pc.get('value'), pc.get('max_scaled_err_10day'))"; }
run config2_f64_b4096 --dtype f64 --batch 4096 --steps 300 --warmup 30
run f64_b65536 --dtype f64 --steps 100 --warmup 10
run f64_b8 --dtype f64 --batch 8 --steps 300 --warmup 30 --no-parity-config
run config5_uncertainty --uncertainty 0.2 --steps 1000 --warmup 100
run b8_f32 --batch 8 --steps 1000 --warmup 100 --no-parity-config
run b4096_f32 --batch 4096 --steps 1000 --warmup 100 --no-parity-config
run b16384_f32 --batch 16384 --steps 1000 --warmup 100 --no-parity-config
run b131072 --batch 131072 --steps 300 --warmup 30 --no-parity-config
run b262144 --batch 262144 --steps 300 --warmup 30 --no-parity-config
run b524288 --batch 524288 --steps 200 --warmup 20 --no-parity-config
run rk4 --scheme rk4 --steps 1000 --warmup 100
run rk3 --scheme rk3 --steps 1000 --warmup 100 --no-parity-config
run rk2 --scheme rk2 --steps 1000 --warmup 100 --no-parity-config
run vecnorm --vecnorm --steps 1000 --warmup 100 --no-parity-config
