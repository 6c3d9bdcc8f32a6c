// vdyn_device.hpp -- device-side math of the 7-DoF planar model and its RK4 step
// for gfx950 (MI355X).  One wavefront lane integrates one rollout; everything
// here lives in registers.
//
// Semantics follow /root/reference/libs/vehicle_model/vehicle_model.py:220-445
// (cited per block below); the arrangement does not.  What is stage-invariant
// in the reference's RK4 (steering sin/cos, normal loads, mu_max, torques: the
// inputs are frozen over the four stages, :429-436) is computed once per step;
// the three divisions and the square root per wheel become one reciprocal and
// one reciprocal square root; the Pacejka split mu_x = s_x mu / s is evaluated
// as s_x * (sin(C atan(B s)) / s) with mu_max folded into the normal load.
//
// Two instantiations of the step exist.  FAST (SAFE = false) is straight-line
// code with no branch at all: a lone wave per SIMD pays tens of cycles for every
// taken branch, and 30-odd "skip the rare fallback" branches per step cost ~17 %
// of the kernel (rocprofv3 SQ_WAIT_ANY, profiles/README.md).  Its trigonometry
// is valid on a bounded range; each lane records whether it stayed inside it,
// and a lane that did not (|yaw| or |delta| > 2^16 rad, a stage yaw increment
// > pi/4) is re-integrated for that step by SAFE (SAFE = true: ROCm device
// library functions, full range) behind one wave-uniform, normally-not-taken
// branch.  A lane's result never depends on what other lanes did.
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_fastmath.hpp"

namespace vdyn {

// Wave-uniform constants, passed by value as a kernel argument (SGPR-resident:
// cheaper than LDS for values every lane shares).  Built on the host in double
// from VdynParams and rounded once to T.
// Per-wheel fit of the Pacejka shape function for the packed fp32 step (vdyn_packed.hpp, pacejka_g2x2):
//   sin(C atan x) / x = c W_C(c),  c = cos(atan x) = 1 / sqrt(1 + x^2),  W_C(c) = sin(C acos c) / sqrt(1 - c^2).
// W_C is the non-integer-C generalisation of the Chebyshev polynomial of the second kind: analytic on (-1, 1], its
// only singularity at c = -1, so ONE polynomial of degree kTireFitDeg in c covers every slip from 0 to infinity
// (c in (0, 1]) -- no argument reduction, no branch, no x > 1 case.  The coefficients depend on C, which belongs to
// the handle: the host fits them when it builds DevParams (make_dev_params, vdyn_kernels.hip) and checks the fp32
// Horner evaluation against double; W[i][wheel], highest degree first.
// fp64 (the trimmed scalar step below and the wheel-parallel one): degree 16 (2.3e-14; round 3's degree 18 held 1.3e-15 and
// cost two more fmas per wheel and stage -- tools/fit_tire_w.py, profiles/r04_tire_fit_degrees.txt), W[i][wheel] as
// well.  Seventeen doubles per wheel fit neither the scalar registers a wave-uniform constant lives in (round 2 kept ONE set, pinned in 34 (then 38)
// VGPRs, and sent handles whose wheels differ in C to the general chain at half the speed) nor, four times over, the
// vector file.  So: wheels that share C (the reference's) -> that one set pinned in VGPRs as in round 2 (pin_tire_fit);
// wheels that differ -> the rollout kernel stages the table in LDS once per workgroup and the step reads it back as
// it goes (fit_horner4_lds: 136 more instructions per step, 0.51 against 0.39 ms on configs[1], against 0.87 for the
// general chain); the wheel-parallel kernel keeps its lane's own wheel's column in registers either way.
// A handle whose fits fail their check takes the general atan -> sine chain (lane_cs).
constexpr int kTireFitDeg = 8;
// Degree of the fp64 fit: 16 (2.3e-14 at the reference's C, at most 3.9e-14 for any accepted C: profiles/
// r05_tire_fit_by_C.txt; the soak test's thousandfold amplifiers then sit at 2e-11 against their 1e-10 bar, a margin of
// five).  -DVDYN_TIRE_FIT_DEG64=18 (an even degree <= 22) builds round 3's fit -- 1.3e-15, two more fmas per wheel and
// stage, vdyn_tire_fit_f64 then fills 19 coefficients -- should that bar ever be tightened.
#ifndef VDYN_TIRE_FIT_DEG64
#define VDYN_TIRE_FIT_DEG64 16
#endif
constexpr int kTireFitDeg64 = VDYN_TIRE_FIT_DEG64;
static_assert(kTireFitDeg64 >= 12 && kTireFitDeg64 <= 22 && kTireFitDeg64 % 2 == 0, "fp64 tire fit degree");
template <typename T> struct TireFit;
template <> struct TireFit<float> {
    float W[kTireFitDeg + 1][4];
};
template <> struct TireFit<double> {
    double W[kTireFitDeg64 + 1][4];
};

// Everything else the step needs of a vehicle.  DevParams = the fits, then this: kernels take a DevParams by value
// as their FIRST argument, so the fits sit at offset 0 of the kernel-argument segment (fit_table_kernarg).
template <typename T>
struct DevCore {
    T inv_m, inv_Izz, inv_Jw;   // 1/m, 1/Izz (vehicle_model.py:376-378), 1/Jw (:379-382)
    T a, b, half_T, rw;         // geometry (:261-271,:378), wheel radius (:284)
    T Fz0F, Fz0R;               // static normal loads (:245-248)
    T DfzxL, DfzxR, DfzyF, DfzyR;  // load-transfer coefficients (:250-253)
    T B[4], C[4];               // Pacejka B, C for FL, FR, RL, RR (:303-306)
    T invB[4];                  // 1/B: 1/(B s) = (1/s)(1/B) feeds atan's |x| > 1 branch for free
    T mu[4];                    // mu_max used by k = 2 controls (drive.py:142: [1,1,1,1])
};
template <typename T>
struct DevParams : TireFit<T>, DevCore<T> {};
__device__ __forceinline__ void pin_tire_fit(DevParams<float> &);
__device__ __forceinline__ void pin_tire_fit(DevParams<double> &P)
{
#pragma unroll
    for (int i = 0; i <= kTireFitDeg64; ++i) asm volatile("" : "+v"(P.W[i][0]));
}

// The fp64 fit table of the running kernel, W[i][wheel] at table[4 i + wheel]: offset 0 of the kernel-argument
// segment (address space 4: constant memory -- a wave-uniform index is a scalar load).
typedef const double __attribute__((address_space(4))) *vdyn_fit_table;
__device__ __forceinline__ vdyn_fit_table fit_table_kernarg()
{
    return (vdyn_fit_table)__builtin_amdgcn_kernarg_segment_ptr();
}

// W_C(c) of the four wheels at once, Horner, fp64: g[k] = sum_i W[i][k] c_k^(16 - i).  The coefficients come from the
// kernel-argument table two degrees (eight doubles = one s_load_dwordx16) at a time; the opaque pointer in front of
// each group keeps the compiler from collecting all 68 of them at the top of the kernel, where they do not fit the
// scalar file (105 v_readlane per step before round 2's fix), while each group's load still goes out ahead of the
// previous group's eight fmas.
// The lane kernels' copy of that table in LDS: 68 doubles, staged once per workgroup (stage_tire_fit), read back by
// every lane at the same address -- a broadcast, no bank conflict.
typedef double __attribute__((address_space(3))) *vdyn_lds_f64;
__device__ __forceinline__ vdyn_lds_f64 fit_table_lds()
{
    __shared__ __align__(16) double s_fit[4 * (kTireFitDeg64 + 1)];
    return (vdyn_lds_f64)s_fit;
}
template <typename T> struct DevParams;
__device__ __forceinline__ void stage_tire_fit(const DevParams<float> &) {}        // fp32: PkParams pins its own pairs
__device__ __forceinline__ void stage_tire_fit(const DevParams<double> &)
{
    const vdyn_lds_f64 t = fit_table_lds();
    const vdyn_fit_table W = fit_table_kernarg();
    for (int i = threadIdx.x; i < 4 * (kTireFitDeg64 + 1); i += blockDim.x) t[i] = W[i];
    __syncthreads();
}

// fp64 handles whose four wheels share C (the reference's do, vehicle_model.py:44-45): column 0 of the table, moved from
// the scalar to the vector registers once, at the top of the kernel.  As wave-uniform kernel arguments the seventeen
// doubles sit in SGPRs, and together with the other constants of the step they overflow the scalar file: the compiler
// then spills SGPRs into VGPR lanes and reads them back one `v_readlane_b32` at a time (105 per RK4 step before this,
// 9 % of the instruction stream).  34 VGPRs (38 at round 3's degree 18) is what the vector file has to spare: one set, not four.
__device__ __forceinline__ void pin_tire_fit(DevParams<float> &) {}
// One C per axle (front tires of one kind, rear tires of another -- the usual way a vehicle's wheels differ): columns
// 0 and 2, both pinned (FITSRC 3).  68 VGPRs; the step still needs no accumulation register and no LDS read.
__device__ __forceinline__ void pin_tire_fit_axles(DevParams<float> &) {}
__device__ __forceinline__ void pin_tire_fit_axles(DevParams<double> &P)
{
#pragma unroll
    for (int i = 0; i <= kTireFitDeg64; ++i) {
        asm volatile("" : "+v"(P.W[i][0]));
        asm volatile("" : "+v"(P.W[i][2]));
    }
}

// W_C(c) of the four wheels at once, Horner, fp64: g[k] = sum_i W[i][k] c_k^(16 - i), the coefficients read from the
// LDS table one degree (four doubles, two 16-byte reads) at a time.  Tried first: scalar loads straight from the
// kernel-argument segment, the coefficient as the fma's scalar operand, two groups in flight -- no VGPR, no LDS, and
// 0.64 ms instead of 0.42 on configs[1]: a scalar-cache hit is ~200 cycles, far more than the eight fp64 fmas it
// was meant to hide behind, and deeper prefetch does not fit the scalar file.  LDS reads land in VGPRs, of which the
// step has a hundred to spare once the pinned set of round 2 is gone, so the compiler runs them several degrees ahead.
typedef double vdyn_d2v __attribute__((ext_vector_type(2)));
template <int kG = 2, int kAhead = 3>     // 2 / 3: 0.558 ms on configs[1] (4 / 1: 0.570, 2 / 6: 0.574, 1 / 6: 0.584)
__device__ __forceinline__ void fit_horner4_lds(const double cc[4], double g[4])
{
    // Groups of kG degrees (2 kG 16-byte reads, 8 kG VGPRs), kAhead groups read ahead of the one being evaluated: a
    // group's reads may not start before all four chains have finished the group kAhead + 1 before it (the asm in
    // front of them takes the chains as inputs).  Left alone, the scheduler issues the 34 reads of an evaluation at
    // once, and the 136 registers they fill push the step into scratch.
    typedef const vdyn_d2v __attribute__((address_space(3))) *lds_ptr;     // stays an LDS address (32 bits), never a flat one
    const lds_ptr t0 = (lds_ptr)fit_table_lds();
    constexpr int kSlots = kAhead + 1;
    constexpr int kRest = kTireFitDeg64 + 1 - 2;                       // coefficients 2 .. kTireFitDeg64
    constexpr int kGroups = kRest / kG;                                // the last group takes the remainder too
    constexpr int kLast = kRest - (kGroups - 1) * kG;
    vdyn_d2v w[kSlots][2 * (kG + kG - 1)];
    auto fetch = [&](int grp, lds_ptr t) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 2 * (grp == kGroups - 1 ? kLast : kG); ++q) w[grp % kSlots][q] = t[2 * (2 + grp * kG) + q];
    };
    lds_ptr t = t0;
    asm volatile("" : "+v"(t) : "v"(cc[0]), "v"(cc[1]), "v"(cc[2]), "v"(cc[3]));
    const vdyn_d2v a0 = t[0], a1 = t[1], b0 = t[2], b1 = t[3];
#pragma unroll
    for (int grp = 0; grp < kAhead && grp < kGroups; ++grp) fetch(grp, t);
    g[0] = ::fma(a0.x, cc[0], b0.x); g[1] = ::fma(a0.y, cc[1], b0.y);
    g[2] = ::fma(a1.x, cc[2], b1.x); g[3] = ::fma(a1.y, cc[3], b1.y);
#pragma unroll
    for (int grp = 0; grp < kGroups; ++grp) {
        if (grp + kAhead < kGroups) {
            lds_ptr tn = t0;
            asm volatile("" : "+v"(tn) : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]));
            fetch(grp + kAhead, tn);
        }
#pragma unroll
        for (int d = 0; d < (grp == kGroups - 1 ? kLast : kG); ++d) {
            const vdyn_d2v w0 = w[grp % kSlots][2 * d], w1 = w[grp % kSlots][2 * d + 1];
            g[0] = ::fma(g[0], cc[0], w0.x); g[1] = ::fma(g[1], cc[1], w0.y);
            g[2] = ::fma(g[2], cc[2], w1.x); g[3] = ::fma(g[3], cc[3], w1.y);
        }
    }
}

// FS of the lane kernels that pin their fit: 1 = the one set the wheels share, 3 = one set per axle.
template <int FS, typename T>
__device__ __forceinline__ void pin_fit(DevParams<T> &P)
{
    if constexpr (FS == 3) pin_tire_fit_axles(P);
    else pin_tire_fit(P);
}

// ---- scalar math, by type and by path ------------------------------------------------
template <typename T, bool SAFE> struct Math;

template <> struct Math<float, false> {
    static constexpr bool kHasRangeLimit = true;
    static constexpr bool kTrim = false;               // the fp32 hot path is the packed step (vdyn_packed.hpp)
    static __device__ __forceinline__ void stage_rot(float, float *, float *) {}
    static constexpr float kStageLimit = 0.0f;
    static __device__ __forceinline__ void sincos_steer(float d, float *s, float *c, bool &ok) { sincos(d, s, c, ok); }
    static __device__ __forceinline__ float rcp(float x) { return fm::rcp(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return fm::rsq(x); }
    // sin(C atan(x)), inv_x = 1/x: the general chain (any C, any sign of x).  CS kernels never get here -- their
    // FAST step is the fitted chain (vdyn_packed.hpp) -- so the flag selects nothing.
    template <bool CS>
    static __device__ __forceinline__ float sin_c_atan(float C, float x, float inv_x)
    {
        return fm::sin_mid(C * fm::atan_rcp(x, inv_x));
    }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c, bool &ok)
    {
        fm::sincos_mid(x, s, c);
        ok = ok && (::fabsf(x) <= fm::kSincosMidLimit);
    }
    // sin, cos of (yaw + d) from sin, cos of yaw: rotate by the small increment
    static __device__ __forceinline__ void stage_sincos(float sy0, float cy0, float, float d, float *s,
                                                        float *c, bool &ok)
    {
        float sd, cd;
        fm::sincos_kernel(d, &sd, &cd);
        *s = ::fmaf(sy0, cd, cy0 * sd);
        *c = ::fmaf(cy0, cd, -sy0 * sd);
        ok = ok && (::fabsf(d) <= fm::kSincosKernelLimit);
    }
};

template <> struct Math<float, true> {
    static constexpr bool kHasRangeLimit = false;
    static constexpr bool kTrim = false;
    static __device__ __forceinline__ void stage_rot(float, float *, float *) {}
    static constexpr float kStageLimit = 0.0f;
    static __device__ __forceinline__ void sincos_steer(float d, float *s, float *c, bool &ok) { sincos(d, s, c, ok); }
    static __device__ __forceinline__ float rcp(float x) { return fm::rcp(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return fm::rsq(x); }
    template <bool CS>
    static __device__ __forceinline__ float sin_c_atan(float C, float x, float) { return ::sinf(C * ::atanf(x)); }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c, bool &) { ::sincosf(x, s, c); }
    static __device__ __forceinline__ void stage_sincos(float, float, float yaw, float, float *s, float *c, bool &)
    {
        ::sincosf(yaw, s, c);
    }
};

template <> struct Math<double, false> {
    static constexpr bool kHasRangeLimit = true;
    static __device__ __forceinline__ double rcp(double x) { return fm64::rcp(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return fm64::rsq(x); }
    template <bool CS>
    static __device__ __forceinline__ double sin_c_atan(double C, double x, double inv_x)
    {
        return fm64::sin_mid(C * fm64::atan_rcp(x, inv_x));       // general chain only: CS takes the fit (tire_force)
    }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c, bool &ok)
    {
        fm64::sincos_mid(x, s, c);
        ok = ok && (::fabs(x) <= fm64::kSincosMidLimit);
    }
    static __device__ __forceinline__ void stage_sincos(double sy0, double cy0, double, double d, double *s,
                                                        double *c, bool &ok)
    {
        double sd, cd;
        fm64::sincos_kernel(d, &sd, &cd);
        *s = ::fma(sy0, cd, cy0 * sd);
        *c = ::fma(cy0, cd, -sy0 * sd);
        ok = ok && (::fabs(d) <= fm64::kSincosKernelLimit);
    }
    // the trimmed fp64 FAST step (kTrim): stages work in the frame of the step's initial yaw
    static constexpr bool kTrim = true;
    static __device__ __forceinline__ void stage_rot(double d, double *s, double *c) { fm64::small_sincos(d, s, c); }
    static constexpr double kStageLimit = fm64::kStageYawLimit64;
    static __device__ __forceinline__ void sincos_steer(double d, double *s, double *c, bool &ok)
    {
        fm64::sincos_kernel(d, s, c);                      // |delta| <= pi/4: no reduction; beyond: SAFE
        ok = ok && (::fabs(d) <= fm64::kSincosKernelLimit);
    }
};

template <> struct Math<double, true> {
    static constexpr bool kHasRangeLimit = false;
    static constexpr bool kTrim = false;
    static __device__ __forceinline__ void stage_rot(double, double *, double *) {}
    static constexpr double kStageLimit = 0.0;
    static __device__ __forceinline__ void sincos_steer(double d, double *s, double *c, bool &ok) { sincos(d, s, c, ok); }
    static __device__ __forceinline__ double rcp(double x) { return fm64::rcp(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return fm64::rsq(x); }
    template <bool CS>
    static __device__ __forceinline__ double sin_c_atan(double C, double x, double) { return ::sin(C * ::atan(x)); }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c, bool &) { ::sincos(x, s, c); }
    static __device__ __forceinline__ void stage_sincos(double, double, double yaw, double, double *s, double *c,
                                                        bool &)
    {
        ::sincos(yaw, s, c);
    }
};

__device__ __forceinline__ float fma_t(float a, float b, float c) { return ::fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return ::fma(a, b, c); }
__device__ __forceinline__ float abs_t(float a) { return ::fabsf(a); }
__device__ __forceinline__ double abs_t(double a) { return ::fabs(a); }
__device__ __forceinline__ float tiny_t(float) { return 1e-30f; }     // sqrt = 1e-15: (B s)^2 below fp32 ulp
__device__ __forceinline__ double tiny_t(double) { return 1e-280; }   // sqrt = 1e-140
__device__ __forceinline__ float steer_limit_t(float) { return fm::kSincosMidLimit; }      // the range Math<T, false>::sincos_steer validates
__device__ __forceinline__ double steer_limit_t(double) { return fm64::kSincosKernelLimit; }
__device__ __forceinline__ float sqrt_t(float a) { return ::sqrtf(a); }
__device__ __forceinline__ double sqrt_t(double a) { return ::sqrt(a); }

// ---- per-step invariants (frozen over the four RK4 stages, :429-436) ----------------
template <typename T>
struct StepInv {
    T cd[4], sd[4];  // cos / sin of the four steering angles (:274-281, :363-373)
    T Fz[4];         // normal loads from the PREVIOUS step's accelerations (:255-258, quirk Q3)
    T muFz[4];       // mu_max_i * Fz_i  (mu_max replaces Pacejka D, :232-235, quirk Q1)
    T tq[4];         // wheel torques (:226)
};

// K2 = true: the drive.py:142-143 pattern -- delta[1] == delta[0], rear angles exactly
// 0, so one sincos serves the front axle and the rear rotations are the identity
// (x*1 + y*0 == x in IEEE) and are skipped.
// PRE (FAST only): sin / cos of the front steering angle arrive precomputed (sd0, cd0).
//   1: evaluated by the same Math<T, false>::sincos when an LDS-shared control table was staged;
//      delta[0] is the angle itself and is range-checked as usual;
//   2: exact (cos = 1 / sqrt(1 + tan^2), sin = tan cos, any angle): delta[0] holds TAN(delta), the form
//      a curvature-driven rollout produces (tan delta = L kappa); the SAFE redo takes the arctangent.
template <typename T, bool K2, bool SAFE, int PRE = 0>
__device__ __forceinline__ void make_step_inv(const DevParams<T> &P, const T delta[4], const T tq[4],
                                              const T mu[4], T ax_prev, T ay_prev, StepInv<T> &c, bool &ok,
                                              T sd0 = T(0), T cd0 = T(1))
{
    using M = Math<T, SAFE>;
    if (PRE != 0 && !SAFE) {
        c.sd[0] = sd0;
        c.cd[0] = cd0;
        if (PRE == 1) ok = ok && (abs_t(delta[0]) <= steer_limit_t(T(0)));
    } else {
        M::sincos_steer(delta[0], &c.sd[0], &c.cd[0], ok);
    }
    if (K2) {
        c.sd[1] = c.sd[0];
        c.cd[1] = c.cd[0];
        c.sd[2] = c.sd[3] = T(0);
        c.cd[2] = c.cd[3] = T(1);
    } else {
        M::sincos_steer(delta[1], &c.sd[1], &c.cd[1], ok);
        M::sincos_steer(delta[2], &c.sd[2], &c.cd[2], ok);
        M::sincos_steer(delta[3], &c.sd[3], &c.cd[3], ok);
    }
    // :255-258
    c.Fz[0] = P.Fz0F - P.DfzxL * ax_prev - P.DfzyF * ay_prev;
    c.Fz[1] = P.Fz0F - P.DfzxR * ax_prev + P.DfzyF * ay_prev;
    c.Fz[2] = P.Fz0R + P.DfzxL * ax_prev - P.DfzyR * ay_prev;
    c.Fz[3] = P.Fz0R + P.DfzxR * ax_prev + P.DfzyR * ay_prev;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c.muFz[i] = mu[i] * c.Fz[i];
        c.tq[i] = tq[i];
    }
}

// One tire: corner velocity (chassis frame) + wheel speed -> tire-frame force
// (fxt, fyt), chassis-frame force (fx, fy) and combined slip s.
//   :274-281 rotation, :284-293 slips (quirk Q4: signed vx for s_x, |vx| for s_y),
//   :296-299 combined slip, :303-348 Pacejka + split, :351-373 forces.
// W (CS, trimmed step only): this wheel's fit coefficients in registers, W[i * ws], highest degree first -- the
// wheel-parallel kernel (one wheel per lane) and the fleet kernel (constants per lane); kernels whose constants are
// wave-uniform evaluate the four wheels together from the kernel-argument table instead (planar_deriv, KARG).
template <typename T, bool STEERED, bool SAFE, bool CS>
__device__ __forceinline__ void tire_force(T B, T invB, T C, const T *W, int ws, T rw, T vxc, T vyc, T w, T cd, T sd,
                                           T muFz, T &fx, T &fy, T &fxt, T &fyt, T &s_out)
{
    using M = Math<T, SAFE>;
    T vx, vy;
    if (STEERED) {
        vx = vxc * cd + vyc * sd;
        vy = vyc * cd - vxc * sd;
    } else {
        vx = vxc;
        vy = vyc;
    }
    if constexpr (M::kTrim && CS) {
        // trimmed form (as vdyn_packed.hpp): slips pre-multiplied by B and
        //   s_x mu / s = (B s_x) G(B s),  G(x) = sin(C atan x) / x = c W_C(c),  c = rsq(1 + x^2)
        // with the handle's own polynomial W (TireFit above): no 1/s, no s, no case x > 1, and quirk Q5
        // (s == 0) needs nothing -- G(0) = C is what the reference's fallback branch returns in the limit
        const T rvxB = M::rcp(vx) * B;
        const T sxb = fma_t(rw, w, -vx) * rvxB;
        const T syb = -vy * abs_t(rvxB);
        const T cc = M::rsqrt(fma_t(sxb, sxb, fma_t(syb, syb, T(1))));
        T g = fma_t(W[0], cc, W[ws]);
#pragma unroll
        for (int i = 2; i <= kTireFitDeg64; ++i) g = fma_t(g, cc, W[i * ws]);
        const T gf = g * (cc * muFz);
        fxt = sxb * gf;
        fyt = syb * gf;
        if (STEERED) {
            fx = fxt * cd - fyt * sd;
            fy = fxt * sd + fyt * cd;
        } else {
            fx = fxt;
            fy = fyt;
        }
        {   // the combined slip itself, diagnostics only (dead code elsewhere): |x| / B, exactly 0 at x = 0
            const T x2 = fma_t(sxb, sxb, syb * syb);
            s_out = x2 * M::rsqrt(x2 > tiny_t(T(0)) ? x2 : tiny_t(T(0))) * invB;
        }
        return;
    }
    const T rvx = M::rcp(vx);
    // rw*w/vx - 1 == (rw*w - vx)/vx; the fused form rounds the small
    // difference once instead of cancelling two O(1) quantities.
    const T sx = fma_t(rw, w, -vx) * rvx;
    const T sy = -vy * abs_t(rvx);
    const T s2 = sx * sx + sy * sy;
    // quirk Q5 (the reference's `s != 0` branch, :309-348), without a branch: clamp
    // s^2 from below by a tiny constant.  For s^2 <= tiny, sin(C atan(B s))/s equals
    // its s -> 0 limit C*B to the last bit, and that is also what the reference's
    // fallback D sin(C atan(B s_x)) evaluates to on an s_x whose square underflowed;
    // with the clamp the regular formula produces exactly that limit (and 0 forces
    // for s_x = s_y = 0) while 1/s stays finite.
    const T s2c = s2 > tiny_t(T(0)) ? s2 : tiny_t(T(0));
    const T rs = M::rsqrt(s2c);
    const T s = s2c * rs;
    const T g = M::template sin_c_atan<CS>(C, B * s, rs * invB) * rs;
    const T gf = g * muFz;
    fxt = sx * gf;
    fyt = sy * gf;
    if (STEERED) {
        fx = fxt * cd - fyt * sd;
        fy = fxt * sd + fyt * cd;
    } else {
        fx = fxt;
        fy = fyt;
    }
    s_out = s2 * rs;  // == s, and exactly 0 when the slip is exactly 0
}

// Diagnostics of one derivative evaluation (vehicle_model.py:420-423 order).
template <typename T>
struct Outputs18 {
    T v[18];
};

// State derivative, vehicle_model.py:220-425.  s[10] = U,V,wz,wFL,wFR,wRL,wRR,yaw,x,y;
// (sy, cy) = sin, cos of s[7].  Returns k[10] and the body accelerations axc, ayc (:413-414).
// FITSRC (fp64 trimmed CS step): 2 = the kernel staged the per-wheel fit table in LDS (stage_tire_fit) and the four
// wheels are evaluated together (fit_horner4_lds); otherwise ONE set, column 0 of P's table -- pinned in VGPRs by the
// kernel (pin_tire_fit) or part of the lane's own constants (fleet).
__device__ __forceinline__ const double *fit_column0(const TireFit<double> &f) { return &f.W[0][0]; }
__device__ __forceinline__ const float *fit_column0(const TireFit<float> &) { return nullptr; }

template <typename T, bool K2, bool DIAG, bool SAFE, bool CS, int FITSRC = 1>
__device__ __forceinline__ void planar_deriv(const DevParams<T> &P, const StepInv<T> &c, const T s[10],
                                             T sy, T cy, T k[10], T &axc, T &ayc, Outputs18<T> *out)
{
    const T U = s[0], V = s[1], wz = s[2];
    // :261-271 (quirk Q8: left wheels at -T/2)
    const T hTw = P.half_T * wz;
    const T vLx = U - hTw, vRx = U + hTw;
    const T vFy = V + P.a * wz, vRy = V - P.b * wz;

    T fx[4], fy[4], fxt[4], fyt[4], sl[4];
    if constexpr (Math<T, SAFE>::kTrim && CS && FITSRC == 2 && sizeof(T) == 8) {
        // the fitted chain of tire_force, the four wheels side by side: slips pre-multiplied by B, c = rsq(1 + x^2),
        // W_C(c) of all four from the kernel-argument table, forces
        using M = Math<T, SAFE>;
        const T vxc[4] = {vLx, vRx, vLx, vRx}, vyc[4] = {vFy, vFy, vRy, vRy};
        T sxb[4], syb[4], cc[4], g[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool steered = i < 2 || !K2;
            const T vx = steered ? vxc[i] * c.cd[i] + vyc[i] * c.sd[i] : vxc[i];
            const T vy = steered ? vyc[i] * c.cd[i] - vxc[i] * c.sd[i] : vyc[i];
            const T rvxB = M::rcp(vx) * P.B[i];
            sxb[i] = fma_t(P.rw, s[3 + i], -vx) * rvxB;
            syb[i] = -vy * abs_t(rvxB);
            cc[i] = M::rsqrt(fma_t(sxb[i], sxb[i], fma_t(syb[i], syb[i], T(1))));
        }
        fit_horner4_lds<2, DIAG ? 1 : 3>(cc, g);      // the diagnostics step has few registers for reads in flight
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool steered = i < 2 || !K2;
            const T gf = g[i] * (cc[i] * c.muFz[i]);
            fxt[i] = sxb[i] * gf;
            fyt[i] = syb[i] * gf;
            fx[i] = steered ? fxt[i] * c.cd[i] - fyt[i] * c.sd[i] : fxt[i];
            fy[i] = steered ? fxt[i] * c.sd[i] + fyt[i] * c.cd[i] : fyt[i];
            if (DIAG) {   // the combined slip itself: |x| / B, exactly 0 at x = 0
                const T x2 = fma_t(sxb[i], sxb[i], syb[i] * syb[i]);
                sl[i] = x2 * M::rsqrt(x2 > tiny_t(T(0)) ? x2 : tiny_t(T(0))) * P.invB[i];
            }
        }
    } else {
        const T *W0 = fit_column0(P);
        const T *WR = FITSRC == 3 && sizeof(T) == 8 ? W0 + 2 : W0;      // 3: the rear axle's own set (pin_tire_fit_axles)
        tire_force<T, true, SAFE, CS>(P.B[0], P.invB[0], P.C[0], W0, 4, P.rw, vLx, vFy, s[3], c.cd[0], c.sd[0],
                                      c.muFz[0], fx[0], fy[0], fxt[0], fyt[0], sl[0]);
        tire_force<T, true, SAFE, CS>(P.B[1], P.invB[1], P.C[1], W0, 4, P.rw, vRx, vFy, s[4], c.cd[1], c.sd[1],
                                      c.muFz[1], fx[1], fy[1], fxt[1], fyt[1], sl[1]);
        tire_force<T, !K2, SAFE, CS>(P.B[2], P.invB[2], P.C[2], WR, 4, P.rw, vLx, vRy, s[5], c.cd[2], c.sd[2],
                                     c.muFz[2], fx[2], fy[2], fxt[2], fyt[2], sl[2]);
        tire_force<T, !K2, SAFE, CS>(P.B[3], P.invB[3], P.C[3], WR, 4, P.rw, vRx, vRy, s[6], c.cd[3], c.sd[3],
                                     c.muFz[3], fx[3], fy[3], fxt[3], fyt[3], sl[3]);
    }

    // :376-385
    T Udot, Vdot;
    if (Math<T, SAFE>::kTrim) {
        // :413-414 first: axc = U_dot - V wz and ayc = V_dot + U wz ARE the force sums over m
        axc = P.inv_m * (fx[0] + fx[1] + fx[2] + fx[3]);
        ayc = P.inv_m * (fy[0] + fy[1] + fy[2] + fy[3]);
        Udot = fma_t(V, wz, axc);
        Vdot = fma_t(-U, wz, ayc);
        const T aI = P.a * P.inv_Izz, bI = P.b * P.inv_Izz, hI = P.half_T * P.inv_Izz;     // loop-invariant
        k[2] = aI * (fy[0] + fy[1]) - bI * (fy[2] + fy[3]) + hI * (fx[1] - fx[0] + fx[3] - fx[2]);
    } else {
        Udot = P.inv_m * (fx[0] + fx[1] + fx[2] + fx[3]) + V * wz;
        Vdot = P.inv_m * (fy[0] + fy[1] + fy[2] + fy[3]) - U * wz;
        k[2] = P.inv_Izz * (P.a * (fy[0] + fy[1]) - P.b * (fy[2] + fy[3])
                            + P.half_T * (fx[1] - fx[0] + fx[3] - fx[2]));
    }
    k[0] = Udot;
    k[1] = Vdot;
    // quirk Q2: front wheels see the tire-frame force, rear wheels the chassis-frame one
    k[3] = (c.tq[0] - P.rw * fxt[0]) * P.inv_Jw;
    k[4] = (c.tq[1] - P.rw * fxt[1]) * P.inv_Jw;
    k[5] = (c.tq[2] - P.rw * fx[2]) * P.inv_Jw;
    k[6] = (c.tq[3] - P.rw * fx[3]) * P.inv_Jw;
    k[7] = wz;
    k[8] = U * cy - V * sy;
    k[9] = U * sy + V * cy;
    if (!Math<T, SAFE>::kTrim) {
        axc = Udot - V * wz;  // :413
        ayc = Vdot + U * wz;  // :414
    }
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out->v[i] = fx[i];
            out->v[4 + i] = fy[i];
            out->v[8 + i] = c.Fz[i];
            out->v[12 + i] = sl[i];
        }
        out->v[16] = fxt[0];
        out->v[17] = fyt[0];
    }
}

// Classic RK4 with frozen inputs, vehicle_model.py:427-445: s[10], (ax, ay) ->
// sn[10], (axn, ayn), the latter the 1-2-2-1 averages of axc, ayc (:442-443).
// DIAG: also state_dot (:440) and the averaged outputs (:441).
// Returns false for a lane that left the validated range of the FAST path.
template <typename T, bool K2, bool DIAG, bool SAFE, bool CS, int PRE = 0, int FITSRC = 1>
__device__ __forceinline__ bool rk4_step(const DevParams<T> &P, const T s[10], T ax, T ay, const T delta[4],
                                         const T tq[4], const T mu[4], T h, T sn[10], T &axn, T &ayn,
                                         T *state_dot, Outputs18<T> *outputs, T sd0 = T(0), T cd0 = T(1))
{
    using M = Math<T, SAFE>;
    bool ok = true;
    StepInv<T> c;
    make_step_inv<T, K2, SAFE, PRE>(P, delta, tq, mu, ax, ay, c, ok, sd0, cd0);
    const T hh = T(0.5) * h;
    T k[10], acc[10], st[10], a1, a2, asx, asy;
    Outputs18<T> o, osum;

    // sin / cos of the stage yaw: one full evaluation per step (quirk Q7: yaw is
    // never wrapped), then stages 2-4 rotate by their increment d = yaw_stage - yaw.
    T sy0, cy0, sy, cy;
    M::sincos(s[7], &sy0, &cy0, ok);

    planar_deriv<T, K2, DIAG, SAFE, CS, FITSRC>(P, c, s, M::kTrim ? T(0) : sy0, M::kTrim ? T(1) : cy0, k, a1, a2, &o);       // K1 (:429)
    asx = a1; asy = a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] = k[i]; st[i] = fma_t(hh, k[i], s[i]); }
    if (DIAG) osum = o;

    // Trimmed FAST step (Math::kTrim): x, y and yaw feed no derivative, so stages 2-4 only need the rotation by
    // their yaw INCREMENT d = yaw_stage - yaw (a short Taylor form, |d| <= 1/32), k[8], k[9] are accumulated in
    // the frame of the initial yaw and the 1-2-2-1 sum is rotated into the global frame once
    // (R(yaw) sum_j w_j R(d_j) u_j == sum_j w_j R(yaw + d_j) u_j).
    T d2 = T(0), d3 = T(0), d4 = T(0);
    if (M::kTrim) { d2 = hh * k[7]; M::stage_rot(d2, &sy, &cy); }
    else M::stage_sincos(sy0, cy0, st[7], hh * k[7], &sy, &cy, ok);
    planar_deriv<T, K2, DIAG, SAFE, CS, FITSRC>(P, c, st, sy, cy, k, a1, a2, &o);        // K2 (:431)
    asx += T(2) * a1; asy += T(2) * a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] = fma_t(T(2), k[i], acc[i]); st[i] = fma_t(hh, k[i], s[i]); }
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 18; ++i) osum.v[i] += T(2) * o.v[i];
    }

    if (M::kTrim) { d3 = hh * k[7]; M::stage_rot(d3, &sy, &cy); }
    else M::stage_sincos(sy0, cy0, st[7], hh * k[7], &sy, &cy, ok);
    planar_deriv<T, K2, DIAG, SAFE, CS, FITSRC>(P, c, st, sy, cy, k, a1, a2, &o);        // K3 (:433)
    asx += T(2) * a1; asy += T(2) * a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] = fma_t(T(2), k[i], acc[i]); st[i] = fma_t(h, k[i], s[i]); }
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 18; ++i) osum.v[i] += T(2) * o.v[i];
    }

    if (M::kTrim) { d4 = h * k[7]; M::stage_rot(d4, &sy, &cy); }
    else M::stage_sincos(sy0, cy0, st[7], h * k[7], &sy, &cy, ok);
    planar_deriv<T, K2, DIAG, SAFE, CS, FITSRC>(P, c, st, sy, cy, k, a1, a2, &o);        // K4 (:435)
    asx += a1; asy += a2;
    if (M::kTrim) {
        T m = abs_t(d2) > abs_t(d3) ? abs_t(d2) : abs_t(d3);
        m = m > abs_t(d4) ? m : abs_t(d4);
        ok = ok && (m <= M::kStageLimit);
    }
    const T h6 = h * T(1.0 / 6.0), sixth = T(1.0 / 6.0);
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] += k[i];
    if (M::kTrim) {                                                          // into the global frame
        const T gx = acc[8] * cy0 - acc[9] * sy0, gy = acc[8] * sy0 + acc[9] * cy0;
        acc[8] = gx;
        acc[9] = gy;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) sn[i] = fma_t(h6, acc[i], s[i]);            // :438
    axn = asx * sixth;                                                       // :442
    ayn = asy * sixth;                                                       // :443
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 10; ++i) state_dot[i] = acc[i] * sixth;                 // :440
#pragma unroll
        for (int i = 0; i < 18; ++i) outputs->v[i] = (osum.v[i] + o.v[i]) * sixth;  // :441
    }
    return ok;
}

// One step for one lane: the FAST path, then SAFE for the lanes that need it.
// The test is wave-uniform (one scalar branch, normally not taken).
//
// fp64 lane kernels without diagnostics (kSelectRedo): behind that branch EVERY lane of the wave runs the SAFE step under
// the full exec mask and each value is then SELECTED per lane -- a lane that stayed in range keeps its FAST result bit
// for bit, the wave pays what a masked redo pays.  Until round 5 the redo was a divergent region everywhere,
// `if (!ok) { SAFE }`, and its join block is where this compiler (ROCm 7.2's clang-22) can place register-allocator
// copies of loop-carried values IN FRONT of the exec restore, i.e. under the redo's partial mask: lanes that did not
// take the redo then keep the previous step's value.  That is what round 4's "RowReader" miscompare was (x, y parked in
// AGPRs, fp64 k = 12), and the shipped fp64 k = 12 shared-table general-chain instance had it too
// (tools/isa/exec_restore_audit.py, tests/test_isa_audit.py).  It takes the register pressure of
// exactly this family -- 512 registers, hundreds of spilled scalars -- so this family no longer HAS the join: a uniform
// branch restores no exec mask.
// Everything else keeps the divergent form, held to "no vector instruction in front of an exec restore at a join" by the
// audit over the built library: the fp32 kernels (measured on one box against round 4's library: the select form cost
// the per-rollout-controls kernel 4.3 % and every fp32 lane kernel 12 VGPRs -- both results of a redone step are alive
// at the select -- for no instance the audit had flagged), the diagnostic instances (the same, times 40 values: 230-340 B
// of scratch instead of 12-20 and +8.7 % on the single-vehicle frame), and the wheel-parallel kernels (the select form
// crashes the compiler there: vdyn_quad.hpp).
__device__ __forceinline__ float atan_lib(float x) { return ::atanf(x); }
__device__ __forceinline__ double atan_lib(double x) { return ::atan(x); }

template <typename T, bool K2, bool DIAG, bool CS, int PRE = 0, int FITSRC = 1>
__device__ __forceinline__ void rk4_advance(const DevParams<T> &P, T s[10], T &ax, T &ay, const T delta[4],
                                            const T tq[4], const T mu[4], T h, T *state_dot,
                                            Outputs18<T> *outputs, T sd0 = T(0), T cd0 = T(1))
{
    T sn[10], axn, ayn;
    const bool ok = rk4_step<T, K2, DIAG, false, CS, PRE, FITSRC>(P, s, ax, ay, delta, tq, mu, h, sn, axn, ayn, state_dot,
                                                                outputs, sd0, cd0);
#ifdef VDYN_MASKED_REDO             // diagnostic build only (tools/isa/reader_all.hip): the divergent form everywhere
    constexpr bool kSelectRedo = false;
#else
    constexpr bool kSelectRedo = !DIAG && sizeof(T) == 8;
#endif
    if (Math<T, false>::kHasRangeLimit) {
        if (__builtin_expect(__any(!ok) != 0, 0)) {
            T dl[4] = {delta[0], delta[1], delta[2], delta[3]};
            if (PRE == 2) {         // delta[0] is the tangent of the (front) steering angle
                dl[0] = dl[1] = atan_lib(delta[0]);
                dl[2] = dl[3] = T(0);
            }
            if constexpr (kSelectRedo) {
                T sr[10], axr, ayr;
                rk4_step<T, K2, false, true, CS>(P, s, ax, ay, dl, tq, mu, h, sr, axr, ayr, nullptr, nullptr);
#pragma unroll
                for (int i = 0; i < 10; ++i) sn[i] = ok ? sn[i] : sr[i];
                axn = ok ? axn : axr;
                ayn = ok ? ayn : ayr;
            } else {
                if (!ok) rk4_step<T, K2, DIAG, true, CS>(P, s, ax, ay, dl, tq, mu, h, sn, axn, ayn, state_dot, outputs);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) s[i] = sn[i];
    ax = axn;
    ay = ayn;
}

}  // namespace vdyn
