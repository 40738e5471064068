#!/usr/bin/env python3
"""Condense a rocprofv3 *_kernel_stats.csv (kernel names can be kilobytes long) into a short table."""
import csv, re, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write("kernel,calls,total_ns,avg_ns,pct,min_ns,max_ns\n")
    for r in rows:
        n = r["Name"]
        m = re.search(r"(step_kernel<[^>]*>|obs_kernel<[^>]*>|reset_kernel<[^>]*>|crop_noise_kernel<[^>]*>|evalf_kernel<[^>]*>|vecnorm_\w+(?:<[^>]*>)?)", n)
        short = m.group(1) if m else re.sub(r"\(.*", "", n)[:80]
        f.write(f"\"{short}\",{r['Calls']},{r['TotalDurationNs']},{float(r['AverageNs']):.0f},{r['Percentage']},{r['MinNs']},{r['MaxNs']}\n")
