// fp32 instantiations, part 2 of 2 (every other launcher): see the end of vdyn_kernels.hip.
#define VDYN_ONLY_F32
#define VDYN_PART 2
#include "vdyn_kernels.hip"
