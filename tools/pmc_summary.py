#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel (short name) mean counter value per dispatch."""
import csv, re, sys, collections, glob
out = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        n = r.get("Kernel_Name", "")
        m = re.search(r"(step_kernel<[^>]*>|obs_kernel<[^>]*>|reset_kernel<[^>]*>)", n)
        if not m: continue
        out[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("kernel,counter,mean_per_dispatch,n_dispatches")
for k in sorted(out):
    for c in sorted(out[k]):
        v = out[k][c]
        print(f"\"{k}\",{c},{sum(v)/len(v):.6g},{len(v)}")
