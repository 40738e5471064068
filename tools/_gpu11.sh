set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "bench_workload_at_full_batch" 2>&1 | grep -E "bench workload|passed|failed"
python bench.py --batch 262144 --steps 3000 --warmup 100 --no-cpu-baseline --no-alt-scheme --no-parity-config 2>/dev/null | grep "^{" > gpurun_out/r05/r05_soak_b262144_bench_line.json
python -c "
import json; d=json.load(open('gpurun_out/r05/r05_soak_b262144_bench_line.json')); print('soak B262144', d['value'], d['roofline']['kernel_ms'], d['integrator_events']['first_attempt_flags'], d['integrator_events']['failed_integrations'], d['integrator_events']['guard_retries'], d['episodes_finished'])"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
