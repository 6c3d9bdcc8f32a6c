#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -S listing.
usage: isa_count.py file.s <substring of mangled kernel name> [--branches]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
key = sys.argv[2]
start = next(i for i, l in enumerate(txt) if l.startswith("_Z") and key in l.split(":")[0])
end = next(i for i in range(start + 1, len(txt)) if txt[i].strip().startswith("s_endpgm"))
body = [l.strip() for l in txt[start + 1:end]]
ins = [(i, l.split()[0]) for i, l in enumerate(body)
       if l and not l.startswith((".", ";")) and not l.split(";")[0].strip().endswith(":")]
print("kernel lines", len(ins))
labels = {l.split(":")[0]: i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
# innermost loops = backward branches
loops = []
for i, l in enumerate(body):
    m = re.match(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"s_branch\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i, m.group(1)))
for lo, hi, name in loops:
    sub = [op for i, op in ins if lo <= i <= hi]
    c = collections.Counter(sub)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    trans = sum(v for k, v in c.items() if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", k))
    print(f"loop {name}: lines {lo}-{hi}  total {len(sub)}  valu {valu}  transcendental {trans}  "
          f"salu {sum(v for k, v in c.items() if k.startswith('s_'))}")
    if "--detail" in sys.argv:
        for k, v in c.most_common(60):
            print(f"    {k:30s}{v}")
