# Evidence for the non-default configurations: for each variant the bench.py JSON line and the
# condensed `rocprofv3 --kernel-trace --stats` summary of the same command, under gpurun_out/<TAG>_variants/ (copy the
# ones to keep into profiles/).   TAG=r02 bash tools/profile_variants.sh
export TMPDIR=/tmp
TAG=${TAG:-r02}
OUT=gpurun_out/${TAG}_variants
mkdir -p $OUT
run() {   # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --no-cpu-baseline "$@" > $OUT/$name.log 2>&1
  grep "^{" $OUT/$name.log > $OUT/${name}_bench_line.json
  f=$(ls $OUT/$name/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/condense_stats.py $f $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/$name
  python - <<PY
import json
d = json.loads(open("$OUT/${name}_bench_line.json").read())
print("$name", "%.3e env-steps/s" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"], d["integrator_events"]["failed_integrations"], d["integrator_events"]["refined_substeps"])
PY
}
run config2_f64_b4096 --dtype f64 --batch 4096 --steps 20 --warmup 3 --no-alt-scheme
run f64_b65536 --dtype f64 --batch 65536 --steps 5 --warmup 1 --no-alt-scheme
run config5_uncertainty --uncertainty 0.2 --steps 200 --warmup 20 --no-alt-scheme
run b524288 --batch 524288 --steps 20 --warmup 3 --no-alt-scheme
run vecnorm --vecnorm --steps 200 --warmup 20 --no-alt-scheme
run rk2 --scheme rk2 --steps 200 --warmup 20 --no-alt-scheme
run rk3 --scheme rk3 --steps 200 --warmup 20 --no-alt-scheme
run b4096_f32 --batch 4096 --steps 50 --warmup 5 --no-alt-scheme
run b8_f32 --batch 8 --steps 50 --warmup 5 --no-alt-scheme --no-obs
