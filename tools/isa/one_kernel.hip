// Quick instruction census of ONE kernel instance (seconds instead of the full translation unit):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp \
//         -I python-motionplanning_amd/csrc -S --cuda-device-only -o /tmp/one.s tools/isa/one_kernel.hip
//   python3 profiles/isa_count.py /tmp/one.s rollout_kernel
#define VDYN_ONLY_F32
#define VDYN_ONLY_F64
#include "vdyn_kernels.hip"
#ifndef ONE_KERNEL
#define ONE_KERNEL rollout_kernel<float, 2, 1, false, true, false>
#endif
namespace vdyn {
template __global__ void ONE_KERNEL(DevParams<float>, int64_t, int, const float *, const float *, const int *, int, int,
                                    float, float *, float *, int, float *, float *);
}
