/*
 * ORACLE -- test infrastructure, NOT product code (see vdyn_oracle.h).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC)
 */
#include <math.h>
#include "vdyn_oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

#define REAL double
#define X(name) name##_f64
#define SIN sin
#define COS cos
#define ATAN atan
#define SQRT sqrt
#define FABS fabs
#define ATAN2 atan2
#define FMOD fmod
#define POW pow
#define FLOOR floor
#include "vdyn_oracle_body.inc"
#undef REAL
#undef X
#undef SIN
#undef COS
#undef ATAN
#undef SQRT
#undef FABS
#undef ATAN2
#undef FMOD
#undef POW
#undef FLOOR

#define REAL float
#define X(name) name##_f32
#define SIN sinf
#define COS cosf
#define ATAN atanf
#define SQRT sqrtf
#define FABS fabsf
#define ATAN2 atan2f
#define FMOD fmodf
#define POW powf
#define FLOOR floorf
#include "vdyn_oracle_body.inc"
