#!/usr/bin/env python3
"""Static look at one kernel of /tmp/glgym_isa.s (tools/isa_summary.py writes it): basic blocks with their loop depth and
instruction counts, largest first -- where the per-window / per-stage instruction budget sits.
    python tools/isa_loops.py <mangled-name-substring> [min_instr]"""
import re, sys
s = open('/tmp/glgym_isa.s').read()
pat = sys.argv[1]; mn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m = re.search(r"^(_ZN\S*" + pat + r"\S*):", s, flags=re.M)
body = s[m.start():]; body = body[:body.index(".Lfunc_end")]
blocks = []; cur = None
for l in body.split("\n"):
    mm = re.match(r"^(\.LBB\d+_\d+|; %bb\.\d+):?\s*(;.*)?$", l)
    if mm or cur is None:
        depth = 0; hdr = ''
        d = re.search(r"Depth=(\d)", l)
        if d: depth = int(d.group(1))
        if 'Loop Header' in l: hdr = 'H'
        cur = {'name': l.split()[0] if l.strip() else 'entry', 'depth': depth, 'hdr': hdr, 'ins': []}
        blocks.append(cur); continue
    d = re.search(r"Loop Header: Depth=(\d)", l)
    if d and not cur['ins']: cur['depth'] = int(d.group(1)); cur['hdr'] = 'H'
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")): cur['ins'].append(t.split()[0])
tot = {}
for b in blocks:
    n = len(b['ins']); v = sum(1 for i in b['ins'] if i.startswith('v_')); tr = sum(1 for i in b['ins'] if re.match(r"v_(exp|log|rcp|sqrt|rsq)_", i))
    pk = sum(1 for i in b['ins'] if i.startswith('v_pk_'))
    b.update(n=n, v=v, tr=tr, pk=pk)
    tot[b['depth']] = tot.get(b['depth'], 0) + n
print("instructions by loop depth:", tot)
for k, b in enumerate(blocks):
    if b['n'] >= mn:
        br = [i for i in b['ins'] if i.startswith(('s_cbranch', 's_branch'))]
        print(f"block #{k:4d} {b['name']:14s} depth {b['depth']}{b['hdr']:1s} instr {b['n']:5d} valu {b['v']:5d} pk {b['pk']:4d} trans {b['tr']:3d} mov {sum(1 for i in b['ins'] if i.startswith(('v_mov','v_accvgpr'))):4d} cndmask {sum(1 for i in b['ins'] if 'cndmask' in i):3d} branches {len(br)}")
