#!/usr/bin/env python3
"""Share of packed (v_pk_*) instructions among the vector instructions of a kernel's innermost (sub-step) loop, from the ISA
tools/isa_summary.py writes to /tmp/glgym_isa.s.  A v_pk_* fp32 op is two lane-operations: it occupies a SIMD for 4 cycles where a
plain op takes 2 (profiles/r02_microbench_issue_rates.txt), and the PMC counts it once -- bench.py's packed-weighted issue-slot
fraction needs the share.
    python tools/isa_summary.py > /dev/null; python tools/pk_share.py [--update profiles/r04_pmc_constants.json]"""
import json, re, sys
s = open('/tmp/glgym_isa.s').read()
KERNELS = {"f32_rk4": "11step_kernelIfLb0ELb1ELb0ELi0ELi1E", "f32_rk3": "11step_kernelIfLb0ELb1ELb0ELi2ELi1E",
           "f32_rk2": "11step_kernelIfLb0ELb1ELb0ELi1ELi1E", "f32_rk4_config5": "11step_kernelIfLb1ELb1ELb0ELi0ELi1E",
           "f32_rk4_quad": "16step_kernel_quadIfLb1ELi0ELb0ELb0E", "f64_rk4_quad": "16step_kernel_quadIdLb0ELi0ELb1ELb0E",
           "f64_rk4_quad_b65536": "16step_kernel_quadIdLb0ELi0ELb1ELb0E"}
out = {}
for variant, pat in KERNELS.items():
    m = re.search(r"^(_ZN\S*" + pat + r"\S*):", s, flags=re.M)
    body = s[m.start():]; body = body[:body.index(".Lfunc_end")]
    depth, valu, pk = 0, {}, {}
    for l in body.split("\n"):
        d = re.search(r"Depth=(\d)", l)
        if d and ("in Loop" in l or "Loop Header" in l):
            depth = int(d.group(1))
        elif re.match(r"^\.LBB\d+_\d+:\s*$", l):
            depth = 0
        t = l.strip()
        if l.startswith("\t") and t.startswith("v_"):
            valu[depth] = valu.get(depth, 0) + 1
            if t.startswith("v_pk_"):
                pk[depth] = pk.get(depth, 0) + 1
    dmax = max(valu)
    out[variant] = {"pk_share_static": pk.get(dmax, 0) / valu[dmax], "loop_depth": dmax, "valu_in_loop": valu[dmax], "pk_in_loop": pk.get(dmax, 0)}
    print(variant, out[variant])
if len(sys.argv) > 2 and sys.argv[1] == "--update":
    d = json.load(open(sys.argv[2]))
    for v, o in out.items():
        if v in d:
            d[v]["pk_share_static"] = round(o["pk_share_static"], 4)
            d[v]["pk_share_source"] = "tools/pk_share.py: v_pk_* / v_* in the sub-step loop of the shipped ISA (%d of %d)" % (o["pk_in_loop"], o["valu_in_loop"])
    json.dump(d, open(sys.argv[2], "w"), indent=1)
