// vdyn_fastmath.hpp -- bounded-range fp32 elementary functions for the Pacejka
// path on gfx950.  The library versions (ocml sinf / atanf / sincosf, IEEE
// division) carry argument reduction for the whole float range, special-case
// branches and a ~10-instruction division expansion; on this path the
// arguments are structurally bounded, so each function below is a short
// straight-line polynomial plus at most one hardware reciprocal.
//
// Accuracy: <= 2 ulp each -- checked at fitting time by tools/fit_polys.py against float64 libm and ON
// THE DEVICE by tests/test_gpu_fastmath.py (dense grids through vdyn_fastmath_eval_*, include/vdyn.h).
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

// sin(y) for y in [0, pi] (Pacejka shape factor 0 <= C <= 2: y = C atan(B s) <= pi):
// reflect about pi/2 instead of the general reduction -- min + 2 adds replace
// mul / rndne / 2 fma / cvt / shift / xor.
__device__ __forceinline__ float sin_0_pi(float y)
{
    const float r = fminf(y, (3.1415927410125732421875f - y) + -8.74227800037248566e-08f);
    const float u = r * r;
    float p = 2.607052693e-06f;
    p = fmaf(p, u, -1.981028618e-04f);
    p = fmaf(p, u, 8.333077654e-03f);
    p = fmaf(p, u, -1.666665971e-01f);
    return fmaf(r * u, p, r);
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

// ---- double precision: same structure, coefficients from tools/fit_polys_f64.py -------
// (Chebyshev interpolants computed in 60-digit arithmetic, 1 ulp each.)  The hardware
// v_rcp_f64 / v_rsq_f64 seeds (2^-24.4 / 2^-24.2 relative on gfx950, tools/ubench/rcp64_probe.hip) are refined
// instead of going through the IEEE division / sqrt expansions (div_scale, div_fmas, div_fixup, ...): by ONE
// cubic step each -- x (1 + e + e^2) with e = 1 - a x; y (1 + e / 2 + 3 e^2 / 8) with e = 1 - a y^2 -- whose
// truncation error e^3 = 2^-73 is far below the rounding.  Measured on 2^20 arguments over 17 decades: 1.00 / 1.24
// ulp, exactly what the two Newton steps of rounds 1-2 gave, in 4 / 6 instructions instead of 5 / 9 (the fp64
// step takes 16 of each per RK4 step: -64 of its 1043 instructions).
namespace fm64 {

__device__ __forceinline__ double rcp(double a)
{
    const double x = __builtin_amdgcn_rcp(a);
    const double e = fma(-a, x, 1.0);
    return fma(x, fma(e, e, e), x);
}

__device__ __forceinline__ double rsq(double a)
{
    const double y = __builtin_amdgcn_rsq(a);
    const double e = fma(-a * y, y, 1.0);
    return fma(y, e * fma(0.375, e, 0.5), y);
}

__device__ __forceinline__ double atan_rcp(double x, double inv_x)
{
    const bool big = fabs(x) > 1.0;
    const double t = big ? inv_x : x;
    const double u = t * t;
    double p = 5.4235777122049743556e-6;
    p = fma(p, u, -0.000068150143017768600815);
    p = fma(p, u, 0.00040788678988090017583);
    p = fma(p, u, -0.0015537533402612332627);
    p = fma(p, u, 0.0042551148264804944648);
    p = fma(p, u, -0.0089974339942772711199);
    p = fma(p, u, 0.015464694610323185545);
    p = fma(p, u, -0.022566714208211488805);
    p = fma(p, u, 0.029121244608854832768);
    p = fma(p, u, -0.034556867496289890707);
    p = fma(p, u, 0.039051162476945009974);
    p = fma(p, u, -0.043181268997216635355);
    p = fma(p, u, 0.047543887590470330271);
    p = fma(p, u, -0.052616416091309294506);
    p = fma(p, u, 0.058821133510041307054);
    p = fma(p, u, -0.066666376799540371802);
    p = fma(p, u, 0.076923050864584886996);
    p = fma(p, u, -0.090909089238135856884);
    p = fma(p, u, 0.11111111103899985381);
    p = fma(p, u, -0.14285714285522452249);
    p = fma(p, u, 0.19999999999997281691);
    p = fma(p, u, -0.33333333333333317957);
    p = fma(p * u, t, t);
    // pi/2 = hi + lo: the subtraction keeps the low word
    return big ? ((copysign(1.570796326794896557998982, x) - p) + copysign(6.12323399573676603586882e-17, x)) : p;
}

// sin(y), |y| up to ~1e9: y = k pi + r, |r| <= pi/2
__device__ __forceinline__ double sin_mid(double y)
{
    const double k = __builtin_rint(y * 0.3183098861837906715377675);
    double r = fma(-k, 3.141592653589793115997963, y);
    r = fma(-k, 1.224646799147353207173764e-16, r);
    const double u = r * r;
    double p = 1.9100730349358852685e-20;
    p = fma(p, u, -8.2181658713046197275e-18);
    p = fma(p, u, 2.8114500908387014114e-15);
    p = fma(p, u, -7.6471636061405583964e-13);
    p = fma(p, u, 1.6059043835456402796e-10);
    p = fma(p, u, -2.5052108385432688701e-8);
    p = fma(p, u, 2.7557319223985856341e-6);
    p = fma(p, u, -0.00019841269841269841204);
    p = fma(p, u, 0.0083333333333333333333);
    p = fma(p, u, -0.16666666666666666667);
    const double s = fma(r * u, p, r);
    const unsigned long long flip = ((unsigned long long)(long long)k) << 63;
    return __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s) ^ flip));
}

__device__ __forceinline__ double sin_0_pi(double y)
{
    const double r = fmin(y, (3.141592653589793115997963 - y) + 1.224646799147353207173764e-16);
    const double u = r * r;
    double p = 1.9100730349358852685e-20;
    p = fma(p, u, -8.2181658713046197275e-18);
    p = fma(p, u, 2.8114500908387014114e-15);
    p = fma(p, u, -7.6471636061405583964e-13);
    p = fma(p, u, 1.6059043835456402796e-10);
    p = fma(p, u, -2.5052108385432688701e-8);
    p = fma(p, u, 2.7557319223985856341e-6);
    p = fma(p, u, -0.00019841269841269841204);
    p = fma(p, u, 0.0083333333333333333333);
    p = fma(p, u, -0.16666666666666666667);
    return fma(r * u, p, r);
}

// (sin d, cos d) of a stage's yaw increment, |d| <= kStageYawLimit64: Taylor to d^7 / d^6
// (2.5e-18 relative / 2.3e-17 absolute at the limit).
constexpr double kStageYawLimit64 = 0.03125;
__device__ __forceinline__ void small_sincos(double d, double *sd, double *cd)
{
    const double u = d * d;
    double ps = fma(u, -1.0 / 5040.0, 1.0 / 120.0);
    ps = fma(ps, u, -1.0 / 6.0);
    *sd = d * fma(ps, u, 1.0);
    double pc = fma(u, -1.0 / 720.0, 1.0 / 24.0);
    pc = fma(pc, u, -0.5);
    *cd = fma(pc, u, 1.0);
}

__device__ __forceinline__ void sincos_kernel(double r, double *sr, double *cr)
{
    const double u = r * r;
    double ps = -7.5865436341019028834e-13;
    ps = fma(ps, u, 1.6058529011960856764e-10);
    ps = fma(ps, u, -2.5052106215978872603e-8);
    ps = fma(ps, u, 2.7557319219291695824e-6);
    ps = fma(ps, u, -0.00019841269841265003859);
    ps = fma(ps, u, 0.0083333333333333314638);
    ps = fma(ps, u, -0.16666666666666666665);
    *sr = fma(r * u, ps, r);
    double pc = 4.7457865678652129218e-14;
    pc = fma(pc, u, -1.1470459438155825967e-11);
    pc = fma(pc, u, 2.0876755781924197192e-9);
    pc = fma(pc, u, -2.7557319221376431924e-7);
    pc = fma(pc, u, 0.000024801587301584612466);
    pc = fma(pc, u, -0.001388888888888888785);
    pc = fma(pc, u, 0.041666666666666666666);
    *cr = fma(u * u, pc, fma(-0.5, u, 1.0));
}

constexpr double kSincosMidLimit = 1073741824.0;  // 2^30
constexpr double kSincosKernelLimit = 0.78539816339744830962;

__device__ __forceinline__ void sincos_mid(double x, double *sn, double *cs)
{
    const double k = __builtin_rint(x * 0.6366197723675813430755351);
    double r = fma(-k, 1.570796326794896557998982, x);
    r = fma(-k, 6.12323399573676603586882e-17, r);
    r = fma(-k, -1.497384904859169832943508e-33, r);
    double sr, cr;
    sincos_kernel(r, &sr, &cr);
    const long long q = (long long)k;
    const bool swap = (q & 1) != 0;
    const double s0 = swap ? cr : sr;
    const double c0 = swap ? sr : cr;
    const unsigned long long fs = ((unsigned long long)(q & 2)) << 62;
    const unsigned long long fc = ((unsigned long long)((q + 1) & 2)) << 62;
    *sn = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(s0) ^ fs));
    *cs = __longlong_as_double((long long)((unsigned long long)__double_as_longlong(c0) ^ fc));
}

}  // namespace fm64
}  // namespace vdyn
