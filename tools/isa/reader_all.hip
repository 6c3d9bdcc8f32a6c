// Reproducer of the code-generation hazard tools/isa/exec_restore_audit.py looks for (12 s to compile):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=iterative-ilp \
//         -I python-motionplanning_amd/csrc -S --cuda-device-only -DVDYN_READER_ALL -DVDYN_MASKED_REDO \
//         -o /tmp/reader_all.s tools/isa/reader_all.hip && python3 tools/isa/exec_restore_audit.py /tmp/reader_all.s
// VDYN_READER_ALL: RowReader for every k and precision (round 4's first form); VDYN_MASKED_REDO: the SAFE redo as the
// divergent region `if (!ok) { ... }` it was until round 5.  Together, in the fp64 k = 12 per-rollout instance, this
// compiler (ROCm 7.2, clang-22) puts four register-allocator copies of the loop-carried x, y (v_accvgpr_write a106..a109)
// in FRONT of the exec restore at the redo's join block: one block flagged.  With either macro left out: none.
#define VDYN_ONLY_F32
#define VDYN_ONLY_F64
#include "vdyn_kernels.hip"
namespace vdyn {
template __global__ void rollout_kernel<double, 12, 0, false, true, false>(DevParams<double>, int64_t, int, const double *,
                                                                          const double *, const int *, int, int, double,
                                                                          double *, double *, int, double *, double *);
}
