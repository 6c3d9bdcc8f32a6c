#!/usr/bin/env python3
"""SGPR-spill census of a translation unit: v_readlane_b32 / v_writelane_b32 (SGPRs spilled into VGPR lanes), scratch
instructions and accumulator moves per kernel, and per loop of the kernels that have any.  How the fp64 kernels'
105 v_readlane per RK4 step were found (`pin_tire_fit`).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize \\
          -mllvm -amdgpu-sched-strategy=iterative-ilp -I python-motionplanning_amd/csrc -S --cuda-device-only \\
          python-motionplanning_amd/csrc/vdyn_kernels_f64_rollout.hip   # (or ..._f64_rest.hip) -o /tmp/f64_all.s        # ~2 min
    python3 tools/isa/spill_census.py /tmp/f64_all.s [kernel-substring]
"""
import os
import re
import subprocess
import sys
from collections import Counter

HERE = os.path.dirname(os.path.abspath(__file__))
path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
txt = open(path).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(txt) if re.match(r"^_ZN4vdyn\w+:", l)]


def demangle(k):
    return subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0]


rows = []
for i, key in starts:
    end = next(j for j in range(i, len(txt)) if txt[j].strip().startswith("s_endpgm"))
    ops = Counter(x.strip().split()[0] for x in txt[i:end] if x.strip() and not x.strip().startswith((";", ".")))
    rows.append((ops["v_readlane_b32"], ops["v_writelane_b32"], sum(v for k, v in ops.items() if k.startswith("scratch_")),
                 key, i))
for rl, wl, sc, key, i in sorted(rows, reverse=True):
    name = demangle(key)
    if want not in name or (rl + wl + sc == 0 and not want):
        continue
    print(f"{rl:6d} readlane {wl:5d} writelane {sc:4d} scratch  {name}")
    if not want:
        continue
    out = subprocess.run([sys.executable, os.path.join(HERE, "..", "..", "profiles", "isa_count.py"), path, key],
                         capture_output=True, text=True).stdout
    for l in out.split("\n"):
        m = re.match(r"loop (\S+): lines (\d+)-(\d+)\s+total (\d+)\s+valu (\d+)", l)
        if not m or int(m.group(5)) < 300:
            continue
        seg = [x.strip() for x in txt[i + int(m.group(2)) - 1:i + int(m.group(3))]]
        c = Counter(x.split()[0] for x in seg if x and not x.startswith((";", ".")))
        print(f"        loop {m.group(1):12s} valu {int(m.group(5)):6d}  readlane {c['v_readlane_b32']:5d}  s_mov {c['s_mov_b32']:4d}")
