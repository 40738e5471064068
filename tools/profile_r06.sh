#!/bin/bash
# Round 6 (same recipe as round 5): PMC constants for every shipped variant of the step kernels (so that no bench line carries roofline.frac = null), the
# kernel-trace summary + bench line of the default workload.  Separate --pmc passes, --kernel-trace only (the guide's recipe).  The
# counters are recorded in the SUSTAINED phase of the workload (60 warm-up steps, 40 profiled) -- not in the first steps after the
# reset, where every exchange law sits on its kink.  Writes gpurun_out/r06_prof/...; copy r06_* into profiles/.
#     bash tools/profile_r06.sh
export TMPDIR=/tmp
OUT=gpurun_out/r06_prof
mkdir -p $OUT
variant() {   # name, kernel-name pattern, n_sub, counter suffix, bench args...
  name=$1; pat=$2; nsub=$3; sfx=$4; shift 4
  if [ -n "$ONLY" ] && [[ ! "$name" =~ $ONLY ]]; then return; fi       # ONLY=regex: re-record a subset of the variants
  V=$OUT/$name; mkdir -p $V
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU_ADD_$sfx SQ_INSTS_VALU_MUL_$sfx SQ_INSTS_VALU_FMA_$sfx SQ_INSTS_VALU_TRANS_$sfx" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $V/pmc$i -- python3 bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-alt-scheme --no-parity --no-parity-config "$@" > $V/pmc$i.log 2>&1 || echo "$name pass $i failed"
  done
  python tools/pmc_summary.py $V/pmc*/*/*counter_collection.csv > $OUT/r06_${name}_pmc_summary.csv
  PMC_KERNEL="$pat" python tools/pmc_summary.py --constants $V/constants.json profiles/r06_${name}_pmc_summary.csv $V/pmc*/*/*counter_collection.csv
  python - <<PY
import csv, glob, json
d = json.load(open("$V/constants.json"))
durs = []
for f in glob.glob("$V/pmc6/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "$pat" in r["Kernel_Name"]:
            durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if durs and "GRBM_GUI_ACTIVE" in d:
    d["kernel_ns_under_pmc"] = sum(durs) / len(durs)
    d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["kernel_ns_under_pmc"]      # the counter is summed over the 8 XCDs
d["n_sub"] = $nsub
d["batch"] = ${PBATCH:-65536}
d["kernel"] = "$pat"
json.dump(d, open("$V/constants.json", "w"), indent=1)
print("$name", {k: d.get(k) for k in ("SQ_INSTS_VALU", "valu_busy", "clock_ghz", "kernel_ns_under_pmc", "traffic_bytes")})
PY
  rm -rf $V/pmc?
}
# default workload: kernel-trace --stats summary + bench line (--no-parity: the accuracy leg launches the same kernel 961 times at B = 64,
# which would mix into the per-kernel average; --no-parity-config: that leg runs the same instantiation at n_sub 192 / window 1)
if [ -z "$ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 300 --warmup 100 --no-parity --no-parity-config > $OUT/bench_stats.log 2>&1
grep "^{" $OUT/bench_stats.log > $OUT/r06_ls5_bench_line.json
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); python tools/condense_stats.py $f $OUT/r06_ls5_bench_kernel_stats.csv; rm -rf $OUT/stats
fi
variant f32_ls5 "step_kernel<float, false, true, false, 3, 1>" 128 F32
variant f32_rk4 "step_kernel<float, false, true, false, 0, 1>" 240 F32 --scheme rk4
variant f32_ls5_config5 "step_kernel<float, true, true, false, 3, 1>" 128 F32 --uncertainty 0.2
# the two-waves-per-SIMD build (what batches of >= 131 072 environments run), recorded at B = 262 144
PBATCH=262144
variant f32_ls5_occ2 "step_kernel<float, false, true, false, 3, 2>" 128 F32 --batch 262144
PBATCH=65536
# fp64 runs the four-lanes-per-environment kernel at every batch size and in every variant (glgym.hip launch_step)
PBATCH=65536
variant f64_ls5_quad_b65536 "step_kernel_quad<double, false, 3, true, false, false>" 128 F64 --dtype f64
# the four-lanes-per-environment kernels (what batches up to 16 384 run), recorded at B = 4 096 (config 2 in fp64)
PBATCH=4096
variant f64_ls5_quad "step_kernel_quad<double, false, 3, true, false, false>" 128 F64 --dtype f64 --batch 4096
variant f64_rk4_quad "step_kernel_quad<double, false, 0, true, false, false>" 240 F64 --dtype f64 --batch 4096 --scheme rk4
variant f32_ls5_quad "step_kernel_quad<float, true, 3, false, false, false>" 128 F32 --batch 4096
PBATCH=65536
python - <<PY
import json
out = {}
for v in ("f32_ls5", "f32_rk4", "f32_ls5_config5", "f32_ls5_occ2", "f64_ls5_quad_b65536", "f64_ls5_quad", "f64_rk4_quad", "f32_ls5_quad"):
    try:
        out[v] = json.load(open("$OUT/%s/constants.json" % v))
    except OSError:
        pass
json.dump(out, open("$OUT/r06_pmc_constants.json", "w"), indent=1)
print(sorted(out))
PY
cat $OUT/r06_ls5_bench_kernel_stats.csv | head -6
