#!/usr/bin/env python3
"""Per-kernel ISA facts of libglgym.so's device code: registers (VGPR + AGPR), scratch, LDS, static instruction counts.
    python tools/isa_summary.py [EXTRA hipcc flags ...]        (cross-compiles, no GPU needed; writes /tmp/glgym_isa.s)"""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
csrc = ROOT / "greenlight-gym2_amd" / "csrc"
out = Path("/tmp/glgym_isa.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize",
                       "--offload-arch=gfx950", "-std=c++17", f"-I{ROOT / 'include'}", "-S", "--cuda-device-only", "-o", str(out),
                       str(csrc / "glgym.hip")] + sys.argv[1:])
s = out.read_text()
dem = {}
for m in re.finditer(r"^\s*\.amdhsa_kernel (\S+)", s, flags=re.M):
    name = m.group(1)
    desc = s[m.start():]; desc = desc[:desc.index(".end_amdhsa_kernel")]
    g = lambda k: int(re.search(r"\.amdhsa_" + k + r" (\d+)", desc).group(1))
    b = re.search(r"^" + re.escape(name) + r":", s, flags=re.M)
    body = s[b.start():]; body = body[:body.index(".Lfunc_end")]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    valu = sum(1 for i in ins if i.startswith("v_")); pk = sum(1 for i in ins if i.startswith("v_pk_"))
    trans = sum(1 for i in ins if re.match(r"v_(exp|log|rcp|sqrt|rsq)_", i))
    acc = int(re.search(r"\.amdhsa_accum_offset (\d+)", desc).group(1))
    nv = g("next_free_vgpr")
    try:
        dn = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except OSError:
        dn = name
    dn = re.sub(r"^void ", "", dn); dn = re.sub(r"\(.*$", "", dn)
    print(f"{dn:<64s} regs {nv:3d} (vgpr {acc:3d} + agpr {nv-acc:3d})  scratch {g('private_segment_fixed_size'):5d} B  lds {g('group_segment_fixed_size'):6d} B  "
          f"static instr {len(ins):6d} valu {valu:6d} pk {pk:4d} trans {trans:4d} readlane {sum(1 for i in ins if i.startswith('v_readlane'))}")
