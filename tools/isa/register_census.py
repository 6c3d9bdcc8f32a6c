#!/usr/bin/env python3
"""Register / scratch census of a translation unit from its `hipcc -S` output: per kernel the private segment
(scratch) bytes, VGPRs, AGPRs, SGPRs and the static spill counts of the code object's metadata.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize \\
          -mllvm -amdgpu-sched-strategy=iterative-ilp -I python-motionplanning_amd/csrc -S --cuda-device-only \\
          python-motionplanning_amd/csrc/vdyn_kernels_f64_rollout.hip   # (or ..._f64_rest.hip) -o /tmp/f64_all.s        # ~2.5 min
    python3 tools/isa/register_census.py /tmp/f64_all.s [kernel-substring] [--all]

Without --all only kernels with scratch or spills are listed.  `vspill` counts VGPRs the allocator parked (in AGPRs
while the private segment is 0); `sspill` SGPRs parked in VGPR lanes (v_writelane / v_readlane: per-loop counts are
tools/isa/spill_census.py's job)."""
import re
import subprocess
import sys

import yaml

args = [a for a in sys.argv[1:] if a != "--all"]
show_all = "--all" in sys.argv
txt = open(args[0]).read()
want = args[1] if len(args) > 1 else ""
md = yaml.safe_load(re.search(r"\.amdgpu_metadata\n(.*?)\n\s*\.end_amdgpu_metadata", txt, re.S).group(1))
rows = [(k[".private_segment_fixed_size"], k[".vgpr_count"], k.get(".agpr_count", 0), k[".sgpr_count"],
         k[".vgpr_spill_count"], k[".sgpr_spill_count"], k[".name"]) for k in md["amdhsa.kernels"]]
names = subprocess.run(["c++filt"], input="\n".join(r[-1] for r in rows), capture_output=True, text=True).stdout.split("\n")
print(f"{'scratch':>7} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'vspill':>6} {'sspill':>6}  kernel")
for r, n in sorted(zip(rows, names), key=lambda x: (-x[0][0], x[1])):
    if want in n and (show_all or r[0] or r[4] or r[5]):
        print(f"{r[0]:7d} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:6d} {r[5]:6d}  {n.split('(')[0][:140]}")
