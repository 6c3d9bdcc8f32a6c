#!/usr/bin/env python3
"""Runs the raw ctypes stub of INTEGRATION.md section 3 verbatim (from the repo root, on the GPU
box) and checks it against KAT-1 of SURVEY.md section 8a: the documentation is executable."""
import re, sys
src = open('INTEGRATION.md').read()
m = re.search(r"## 3\. Raw `ctypes` stub.*?```python\n(.*?)```", src, re.S)
code = m.group(1)
import torch  # noqa
ns = {}
exec(code, ns)
out = ns['planar_model_RK4']([25.0,0,0]+[25.0/0.308309813617345]*4+[0,0,0], [0]*4, [1.0]*4, [0.02,0.02,0,0], 0.0, 0.0, 1e-4)
print(out[0][:3], out[7], out[8])
assert abs(out[8] - 2.944897222404597) < 1e-9 and abs(out[7] + 0.030200889513079934) < 1e-9
print("INTEGRATION.md stub ok (KAT-1)")
