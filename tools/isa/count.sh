#!/bin/bash
# usage: tools/isa/count.sh [kernel-substring]   (static VALU census of the headline kernel's step loop)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
  -I python-motionplanning_amd/csrc -S --cuda-device-only -o /tmp/one.s tools/isa/one_kernel.hip 2>&1 | grep -v "hip-link" || true
python3 profiles/isa_count.py /tmp/one.s ${1:-rollout_kernel} | awk '{for(i=1;i<=NF;i++) if($i=="valu" && $(i+1)>100) print}'
grep -E "vgpr_count|sgpr_count|scratch|NumVgprs|ScratchSize" /tmp/one.s | head -6
