// fp32 instantiations of every kernel and launcher (see vdyn_kernels.hip).
#define VDYN_ONLY_F32
#include "vdyn_kernels.hip"
