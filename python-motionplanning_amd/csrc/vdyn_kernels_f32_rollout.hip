// fp32 instantiations, part 1 of 2 (the rollout launcher: lane and wheel-parallel kernels): see the end of vdyn_kernels.hip.
#define VDYN_ONLY_F32
#define VDYN_PART 1
#include "vdyn_kernels.hip"
