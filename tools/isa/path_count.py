#!/usr/bin/env python3
"""VALU instructions between two labels of a kernel listing, per basic block.
usage: path_count.py file.s <kernel substring> <first label> <last line: label or 'branch-to-first'>"""
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start + 1, len(txt)) if txt[i].strip().startswith("s_endpgm"))
body = [l.strip() for l in txt[start + 1:end]]
first = sys.argv[3]
i0 = next(i for i, l in enumerate(body) if l.startswith(first + ":"))
blocks, cur, name = [], [], first
for i in range(i0 + 1, len(body)):
    l = body[i]
    if not l or l.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", l):
        continue
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur))
        name, cur = m.group(1), []
        continue
    cur.append(l)
    if re.match(r"s_branch\s+" + re.escape(first) + r"\b", l):
        blocks.append((name, cur))
        break
for name, ins in blocks:
    valu = sum(1 for x in ins if x.startswith("v_"))
    mov = sum(1 for x in ins if x.startswith("v_mov_b32"))
    trans = sum(1 for x in ins if re.match(r"v_(rcp|rsq|sqrt)_", x))
    br = [x for x in ins if x.startswith(("s_cbranch", "s_branch"))]
    print(f"{name:12s} insts {len(ins):5d} valu {valu:5d} (mov {mov:3d}, trans {trans:3d})  {'; '.join(br)}")
