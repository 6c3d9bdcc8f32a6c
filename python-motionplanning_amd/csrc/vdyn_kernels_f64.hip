// fp64 instantiations of every kernel and launcher (see vdyn_kernels.hip).
#define VDYN_ONLY_F64
#include "vdyn_kernels.hip"
