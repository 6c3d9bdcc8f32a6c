// vdyn_fastmath.hpp -- bounded-range fp32 elementary functions for the Pacejka
// path on gfx950.  The library versions (ocml sinf / atanf / sincosf, IEEE
// division) carry argument reduction for the whole float range, special-case
// branches and a ~10-instruction division expansion; on this path the
// arguments are structurally bounded, so each function below is a short
// straight-line polynomial plus at most one hardware reciprocal.
//
// Accuracy (checked by tools/fit_polys.py against float64 libm, and on the GPU
// by tests/test_gpu_parity.py::test_fastmath_accuracy): <= 2 ulp each.
// Coefficients: tools/fit_polys.py (Lawson minimax fits, rounded to float).
#pragma once
#include <hip/hip_runtime.h>

namespace vdyn {
namespace fm {

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32, 1 ulp
__device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }    // v_rsq_f32, 1 ulp

// atan(x), given inv_x = 1/x (the caller has it for free; x = B s >= 0 on this path,
// but the sign is honoured).  atan(t) = t P(t^2) on [-1,1]; |x| > 1: atan(x) =
// copysign(pi/2, x) - atan(1/x).
__device__ __forceinline__ float atan_rcp(float x, float inv_x)
{
    const bool big = fabsf(x) > 1.0f;
    const float t = big ? inv_x : x;
    const float u = t * t;
    float p = 2.872858429e-03f;
    p = fmaf(p, u, -1.616817340e-02f);
    p = fmaf(p, u, 4.286647215e-02f);
    p = fmaf(p, u, -7.520283014e-02f);
    p = fmaf(p, u, 1.064901948e-01f);
    p = fmaf(p, u, -1.420586258e-01f);
    p = fmaf(p, u, 1.999291778e-01f);
    p = fmaf(p, u, -3.333308995e-01f);
    p = fmaf(p * u, t, t);  // t + t u P'(u): the leading coefficient is exactly 1
    return big ? (copysignf(1.57079637050628662109375f, x) - p) : p;
}

// sin(y) for |y| up to ~1e4: y = k pi + r, |r| <= pi/2, sin y = (-1)^k sin r.
__device__ __forceinline__ float sin_mid(float y)
{
    const float k = __builtin_rintf(y * 0.318309886183790671538f);
    float r = fmaf(-k, 3.1415927410125732421875f, y);        // float(pi)
    r = fmaf(-k, -8.74227800037248566e-08f, r);              // pi - float(pi)
    const float u = r * r;
    float p = 2.607052693e-06f;
    p = fmaf(p, u, -1.981028618e-04f);
    p = fmaf(p, u, 8.333077654e-03f);
    p = fmaf(p, u, -1.666665971e-01f);
    const float s = fmaf(r * u, p, r);
    const unsigned flip = ((unsigned)(int)k) << 31;
    return __uint_as_float(__float_as_uint(s) ^ flip);
}

// sin / cos kernels on |r| <= pi/4 (1 ulp)
__device__ __forceinline__ void sincos_kernel(float r, float *sr, float *cr)
{
    const float u = r * r;
    float ps = -1.951163867e-04f;
    ps = fmaf(ps, u, 8.332134224e-03f);
    ps = fmaf(ps, u, -1.666665375e-01f);
    *sr = fmaf(r * u, ps, r);
    float pc = 2.443367339e-05f;
    pc = fmaf(pc, u, -1.388732577e-03f);
    pc = fmaf(pc, u, 4.166664556e-02f);
    *cr = fmaf(u * u, pc, fmaf(-0.5f, u, 1.0f));
}

// Largest |x| for which sincos_mid's three-term Cody-Waite reduction holds 1-2 ulp.
constexpr float kSincosMidLimit = 65536.0f;
// Largest stage increment the kernels above cover without reduction.
constexpr float kSincosKernelLimit = 0.78539816339f;

// sin and cos for |x| <= kSincosMidLimit, straight-line (yaw is never wrapped by
// the reference -- quirk Q7 -- so this has to take angles well beyond 2 pi).
__device__ __forceinline__ void sincos_mid(float x, float *sn, float *cs)
{
    const float k = __builtin_rintf(x * 0.636619772367581343076f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);       // float(pi/2)
    r = fmaf(-k, -4.37113900018624283e-08f, r);              // pi/2 - float(pi/2)
    r = fmaf(-k, -1.7151245100059e-15f, r);
    float sr, cr;
    sincos_kernel(r, &sr, &cr);
    const int q = (int)k;
    const bool swap = (q & 1) != 0;
    const float s0 = swap ? cr : sr;
    const float c0 = swap ? sr : cr;
    const unsigned fs = ((unsigned)(q & 2)) << 30;           // quadrants 2, 3: sin < 0
    const unsigned fc = ((unsigned)((q + 1) & 2)) << 30;     // quadrants 1, 2: cos < 0
    *sn = __uint_as_float(__float_as_uint(s0) ^ fs);
    *cs = __uint_as_float(__float_as_uint(c0) ^ fc);
}

}  // namespace fm
}  // namespace vdyn
