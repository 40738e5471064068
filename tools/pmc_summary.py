#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (short name) mean counter value per dispatch."""
import csv, re, sys, collections, glob
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ([] if (len(sys.argv) > 1 and sys.argv[1] == "--constants") else sys.argv[1:]):
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        m = re.search(r"(step_kernel(?:_quad)?<[^>]*>|evalf_kernel(?:_quad)?<[^>]*>|obs_kernel<[^>]*>|reset_kernel<[^>]*>)", n)
        if not m: continue
        out[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not (len(sys.argv) > 1 and sys.argv[1] == "--constants"):
    print("kernel,counter,mean_per_dispatch,n_dispatches")
for k in sorted(out):
    for c in sorted(out[k]):
        v = out[k][c]
        print(f"\"{k}\",{c},{sum(v)/len(v):.6g},{len(v)}")

# Constants file bench.py reads for its roofline block: per-launch counters of the default fp32 RK4 step kernel.
#   python tools/pmc_summary.py --constants OUT.json SOURCE_LABEL  file.csv ...
if len(sys.argv) > 1 and sys.argv[1] == "--constants":
    import json
    dst, label, files = sys.argv[2], sys.argv[3], sys.argv[4:]
    import os
    pattern = os.environ.get("PMC_KERNEL", "step_kernel<float, false, true, false, 0")      # which instantiation (tools/profile_r03.sh)
    vals = collections.defaultdict(list)
    dur = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if pattern not in r.get("Kernel_Name", ""):
                continue
            vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in vals.items()}
    out = {k: c[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32",
                             "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64",
                             "SQ_INSTS_VALU_ADD_F64", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES",
                             "FETCH_SIZE", "WRITE_SIZE", "GRBM_GUI_ACTIVE") if k in c}
    # gfx950: FETCH_SIZE reports half of the fetched bytes (MI355X_MICROARCH.md, HBM section); both counters are in KiB
    out["traffic_bytes"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
    out["valu_busy"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    out["source"] = label
    json.dump(out, open(dst, "w"), indent=1)
