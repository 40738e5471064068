#!/usr/bin/env python3
"""Registers / scratch / LDS / static instruction counts of the step and evalF kernels from an existing device assembly
(`make -C greenlight-gym2_amd/csrc asm` writes /tmp/glgym.s).   python tools/isa_table.py [/tmp/glgym.s] [filter]"""
import re, subprocess, sys
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/glgym.s"
flt = sys.argv[2] if len(sys.argv) > 2 else "step_kernel|evalf"
s = open(path).read()
rows = []
for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", s, flags=re.M):
    name = m.group(1)
    desc = s[m.start():]; desc = desc[:desc.index(".end_amdhsa_kernel")]
    g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", desc).group(1))
    b = re.search(r"^" + re.escape(name) + r":", s, flags=re.M)
    body = s[b.start():]; body = body[:body.index(".Lfunc_end")]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1)); nv = g("next_free_vgpr")
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace("(anonymous namespace)::", ""); dn = re.sub(r"^void ", "", dn); dn = re.sub(r"\(.*$", "", dn)
    if re.search(flt, dn):
        rows.append((dn, nv, acc, nv - acc, g('private_segment_fixed_size'), g('group_segment_fixed_size'), len(ins), sum(1 for i in ins if i.startswith('v_pk_'))))
for r in sorted(rows):
    print("%-66s regs %3d (v %3d a %3d) scratch %5d lds %6d instr %6d pk %4d" % r)
