export TMPDIR=/tmp
set -x
SCHEME=${SCHEME:-rk4}
OUT=gpurun_out/${TAG:-v9}_$SCHEME
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 20 --warmup 3 --scheme $SCHEME > $OUT/bench_stats.log 2>&1
grep "^{" $OUT/bench_stats.log > $OUT/bench_line.json
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc$i -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt-scheme --scheme $SCHEME > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
f=$(ls $OUT/stats/*/*kernel_stats.csv | head -1); python tools/condense_stats.py $f $OUT/kernel_stats.csv
python tools/pmc_summary.py $OUT/pmc*/*/*counter_collection.csv > $OUT/pmc_summary.csv
# kernel duration under the counter passes -> clock = GRBM_GUI_ACTIVE / duration; constants for bench.py's roofline block
python - <<PY
import csv, glob, json, subprocess, sys
subprocess.check_call([sys.executable, "tools/pmc_summary.py", "--constants", "$OUT/pmc_constants.json",
                       "profiles/${TAG:-v9}_${SCHEME}_pmc_summary.csv"] + glob.glob("$OUT/pmc*/*/*counter_collection.csv"))
d = json.load(open("$OUT/pmc_constants.json"))
durs = []
for f in glob.glob("$OUT/pmc7/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "step_kernel<float, false, true, false, 0" in r["Kernel_Name"]:
            durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
if durs and "GRBM_GUI_ACTIVE" in d:
    d["kernel_ns_under_pmc"] = sum(durs) / len(durs)
    d["clock_ghz"] = d["GRBM_GUI_ACTIVE"] / 8.0 / d["kernel_ns_under_pmc"]      # the counter is summed over the 8 XCDs
json.dump(d, open("$OUT/pmc_constants.json", "w"), indent=1)
print(d)
PY
cat $OUT/kernel_stats.csv | head -8
grep step_kernel $OUT/pmc_summary.csv
