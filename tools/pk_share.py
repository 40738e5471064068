#!/usr/bin/env python3
"""Share of packed (v_pk_*) instructions among the vector instructions of a kernel's innermost (sub-step) loop, and the static
instruction counts that tie a set of recorded PMC constants to the ISA they were recorded on.

A v_pk_* fp32 op is two lane-operations: it occupies a SIMD for 4 cycles where a plain op takes 2
(profiles/r02_microbench_issue_rates.txt), and the PMC counts it once -- bench.py's packed-weighted issue-slot fraction needs the
share.  The fingerprint (`isa_instr_total`, `isa_valu_in_loop`, `isa_pk_in_loop`) is stored next to the counters in
profiles/rNN_pmc_constants.json; tests/test_capi_surface.py recompiles the device code and fails when the shipped kernels no
longer are the ones the counters were recorded on.

    make -C greenlight-gym2_amd/csrc asm            # writes /tmp/glgym.s with the Makefile's flags
    python tools/pk_share.py [--update profiles/r06_pmc_constants.json] [ASM]"""
import json, re, sys

# variant name (bench.py's) -> mangled-name fragment of the instantiation it runs
KERNELS = {"f32_ls5": "11step_kernelIfLb0ELb1ELb0ELi3ELi1E", "f32_rk4": "11step_kernelIfLb0ELb1ELb0ELi0ELi1E",
           "f32_rk3": "11step_kernelIfLb0ELb1ELb0ELi2ELi1E", "f32_rk2": "11step_kernelIfLb0ELb1ELb0ELi1ELi1E",
           "f32_ls5_config5": "11step_kernelIfLb1ELb1ELb0ELi3ELi1E", "f32_rk4_config5": "11step_kernelIfLb1ELb1ELb0ELi0ELi1E",
           "f32_ls5_quad": "16step_kernel_quadIfLb1ELi3ELb0ELb0ELb0E", "f32_rk4_quad": "16step_kernel_quadIfLb1ELi0ELb0ELb0ELb0E",
           "f64_ls5_quad": "16step_kernel_quadIdLb0ELi3ELb1ELb0ELb0E", "f64_ls5_quad_b65536": "16step_kernel_quadIdLb0ELi3ELb1ELb0ELb0E",
           "f64_rk4_quad": "16step_kernel_quadIdLb0ELi0ELb1ELb0ELb0E", "f64_rk4_quad_b65536": "16step_kernel_quadIdLb0ELi0ELb1ELb0ELb0E",
           "f32_ls5_occ2": "11step_kernelIfLb0ELb1ELb0ELi3ELi2E"}


def loop_stats(asm: str, fragment: str):
    """-> {"isa_instr_total", "isa_valu_in_loop", "isa_pk_in_loop", "pk_share_static", "loop_depth"} of the kernel whose mangled
    name contains `fragment`, or None.  "Loop" = the deepest loop nest the compiler annotates (the sub-step loop)."""
    m = re.search(r"^(_ZN\S*" + fragment + r"\S*):", asm, flags=re.M)
    if not m:
        return None
    body = asm[m.start():]; body = body[:body.index(".Lfunc_end")]
    depth, valu, pk, total = 0, {}, {}, 0
    for l in body.split("\n"):
        d = re.search(r"Depth=(\d)", l)
        if d and ("in Loop" in l or "Loop Header" in l):
            depth = int(d.group(1))
        elif re.match(r"^\.LBB\d+_\d+:\s*$", l):
            depth = 0
        t = l.strip()
        if l.startswith("\t") and not t.startswith((".", ";")):
            total += 1
            if t.startswith("v_"):
                valu[depth] = valu.get(depth, 0) + 1
                if t.startswith("v_pk_"):
                    pk[depth] = pk.get(depth, 0) + 1
    dmax = max(valu)
    return {"isa_instr_total": total, "isa_valu_in_loop": valu[dmax], "isa_pk_in_loop": pk.get(dmax, 0),
            "pk_share_static": round(pk.get(dmax, 0) / valu[dmax], 4), "loop_depth": dmax}


if __name__ == "__main__":
    args = sys.argv[1:]
    upd = None
    if args and args[0] == "--update":
        upd, args = args[1], args[2:]
    asm = open(args[0] if args else "/tmp/glgym.s").read()
    out = {v: loop_stats(asm, frag) for v, frag in KERNELS.items()}
    for v, o in out.items():
        print(v, o)
    if upd:
        d = json.load(open(upd))
        for v, o in out.items():
            if v in d and o:
                d[v].update({k: o[k] for k in ("pk_share_static", "isa_instr_total", "isa_valu_in_loop", "isa_pk_in_loop")})
                d[v]["isa_fragment"] = KERNELS[v]
                d[v]["pk_share_source"] = ("tools/pk_share.py: v_pk_* / v_* in the sub-step loop of the shipped ISA (%d of %d)"
                                           % (o["isa_pk_in_loop"], o["isa_valu_in_loop"]))
        json.dump(d, open(upd, "w"), indent=1)
