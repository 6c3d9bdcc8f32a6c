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
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_fastmath.hpp"

namespace vdyn {

// Wave-uniform constants, passed by value as a kernel argument (SGPR-resident:
// cheaper than LDS for values every lane shares).  Built on the host in double
// from VdynParams and rounded once to T.
template <typename T>
struct DevParams {
    T inv_m, inv_Izz, inv_Jw;   // 1/m, 1/Izz (vehicle_model.py:376-378), 1/Jw (:379-382)
    T a, b, half_T, rw;         // geometry (:261-271,:378), wheel radius (:284)
    T Fz0F, Fz0R;               // static normal loads (:245-248)
    T DfzxL, DfzxR, DfzyF, DfzyR;  // load-transfer coefficients (:250-253)
    T B[4], C[4];               // Pacejka B, C for FL, FR, RL, RR (:303-306)
    T invB[4];                  // 1/B: 1/(B s) = (1/s)(1/B) feeds atan's x > 1 branch for free
    T mu[4];                    // mu_max used by k = 2 controls (drive.py:142: [1,1,1,1])
};

// ---- scalar math wrappers -------------------------------------------------------
template <typename T> struct Math;

template <> struct Math<float> {
    // bounded-range straight-line versions (vdyn_fastmath.hpp)
    static __device__ __forceinline__ float sin_pacejka(float y) { return fm::sin_mid(y); }
    static __device__ __forceinline__ float atan_pos(float x, float inv_x) { return fm::atan_pos(x, inv_x); }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c) { fm::sincos_any(x, s, c); }
    static __device__ __forceinline__ float rcp(float x) { return fm::rcp(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return fm::rsq(x); }
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return ::fmaf(a, b, c); }
    static __device__ __forceinline__ float abs(float x) { return ::fabsf(x); }
    // sin d, cos d for a small stage increment d of the yaw angle; `ok` = series valid
    static __device__ __forceinline__ bool small_sincos(float d, float *s, float *c)
    {
        const float u = d * d;
        *s = fmaf(d * u, fmaf(u, fmaf(u, -1.0f / 5040.0f, 1.0f / 120.0f), -1.0f / 6.0f), d);
        *c = fmaf(u, fmaf(u, fmaf(u, fmaf(u, 1.0f / 40320.0f, -1.0f / 720.0f), 1.0f / 24.0f), -0.5f), 1.0f);
        return ::fabsf(d) <= 0.25f;
    }
};

template <> struct Math<double> {
    static __device__ __forceinline__ double sin_pacejka(double y) { return ::sin(y); }
    static __device__ __forceinline__ double atan_pos(double x, double) { return ::atan(x); }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return ::fma(a, b, c); }
    static __device__ __forceinline__ double abs(double x) { return ::fabs(x); }
    static __device__ __forceinline__ bool small_sincos(double d, double *s, double *c)
    {
        const double u = d * d;
        double ps = -1.0 / 39916800.0;                 // d^11 / 11!
        ps = ::fma(ps, u, 1.0 / 362880.0);
        ps = ::fma(ps, u, -1.0 / 5040.0);
        ps = ::fma(ps, u, 1.0 / 120.0);
        ps = ::fma(ps, u, -1.0 / 6.0);
        *s = ::fma(d * u, ps, d);
        double pc = 1.0 / 479001600.0;                 // d^12 / 12!
        pc = ::fma(pc, u, -1.0 / 3628800.0);
        pc = ::fma(pc, u, 1.0 / 40320.0);
        pc = ::fma(pc, u, -1.0 / 720.0);
        pc = ::fma(pc, u, 1.0 / 24.0);
        pc = ::fma(pc, u, -0.5);
        *c = ::fma(pc, u, 1.0);
        return ::fabs(d) <= 0.0625;
    }
};

// ---- per-step invariants (frozen over the four RK4 stages, :429-436) ----------------
template <typename T>
struct StepInv {
    T cd[4], sd[4];  // cos / sin of the four steering angles (:274-281, :363-373)
    T Fz[4];         // normal loads from the PREVIOUS step's accelerations (:255-258, quirk Q3)
    T muFz[4];       // mu_max_i * Fz_i  (mu_max replaces Pacejka D, :232-235, quirk Q1)
    T tq[4];         // wheel torques (:226)
};

// REAR = false: rear steering angles are exactly 0 (k = 2 controls), so the
// rear rotations are the identity and are skipped (x*1 + y*0 == x in IEEE).
template <typename T, bool REAR>
__device__ __forceinline__ void make_step_inv(const DevParams<T> &P, const T delta[4], const T tq[4],
                                              const T mu[4], T ax_prev, T ay_prev, StepInv<T> &c)
{
    using M = Math<T>;
    M::sincos(delta[0], &c.sd[0], &c.cd[0]);
    if (delta[1] == delta[0]) { c.sd[1] = c.sd[0]; c.cd[1] = c.cd[0]; }
    else M::sincos(delta[1], &c.sd[1], &c.cd[1]);
    if (REAR) {
        M::sincos(delta[2], &c.sd[2], &c.cd[2]);
        M::sincos(delta[3], &c.sd[3], &c.cd[3]);
    } else {
        c.sd[2] = c.sd[3] = T(0);
        c.cd[2] = c.cd[3] = T(1);
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
template <typename T, bool STEERED>
__device__ __forceinline__ void tire_force(T B, T invB, T C, T rw, T vxc, T vyc, T w, T cd, T sd, T muFz,
                                           T &fx, T &fy, T &fxt, T &fyt, T &s_out)
{
    using M = Math<T>;
    T vx, vy;
    if (STEERED) {
        vx = vxc * cd + vyc * sd;
        vy = vyc * cd - vxc * sd;
    } else {
        vx = vxc;
        vy = vyc;
    }
    const T rvx = M::rcp(vx);
    // rw*w/vx - 1 == (rw*w - vx)/vx; the fused form rounds the small
    // difference once instead of cancelling two O(1) quantities.
    const T sx = M::fma(rw, w, -vx) * rvx;
    const T sy = -vy * M::abs(rvx);
    const T s2 = sx * sx + sy * sy;
    const T rs = M::rsqrt(s2);
    const T s = s2 * rs;
    // quirk Q5: when s == 0 the reference evaluates D sin(C atan(B s_x)) on an
    // s_x whose square underflowed; sin(C atan(B e)) == C B e to the last bit
    // for such e, so the s -> 0 limit C*B of sin(C atan(B s))/s is exact there.
    const T g = (s2 == T(0)) ? C * B : M::sin_pacejka(C * M::atan_pos(B * s, rs * invB)) * rs;
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
    s_out = (s2 == T(0)) ? T(0) : s;
}

// Diagnostics of one derivative evaluation (vehicle_model.py:420-423 order).
template <typename T>
struct Outputs18 {
    T v[18];
};

// State derivative, vehicle_model.py:220-425.  s[10] = U,V,wz,wFL,wFR,wRL,wRR,yaw,x,y;
// (sy, cy) = sin, cos of s[7].  Returns k[10] and the body accelerations axc, ayc (:413-414).
template <typename T, bool REAR, bool DIAG>
__device__ __forceinline__ void planar_deriv(const DevParams<T> &P, const StepInv<T> &c, const T s[10],
                                             T sy, T cy, T k[10], T &axc, T &ayc, Outputs18<T> *out)
{
    const T U = s[0], V = s[1], wz = s[2];
    // :261-271 (quirk Q8: left wheels at -T/2)
    const T hTw = P.half_T * wz;
    const T vLx = U - hTw, vRx = U + hTw;
    const T vFy = V + P.a * wz, vRy = V - P.b * wz;

    T fx[4], fy[4], fxt[4], fyt[4], sl[4];
    tire_force<T, true>(P.B[0], P.invB[0], P.C[0], P.rw, vLx, vFy, s[3], c.cd[0], c.sd[0], c.muFz[0],
                        fx[0], fy[0], fxt[0], fyt[0], sl[0]);
    tire_force<T, true>(P.B[1], P.invB[1], P.C[1], P.rw, vRx, vFy, s[4], c.cd[1], c.sd[1], c.muFz[1],
                        fx[1], fy[1], fxt[1], fyt[1], sl[1]);
    tire_force<T, REAR>(P.B[2], P.invB[2], P.C[2], P.rw, vLx, vRy, s[5], c.cd[2], c.sd[2], c.muFz[2],
                        fx[2], fy[2], fxt[2], fyt[2], sl[2]);
    tire_force<T, REAR>(P.B[3], P.invB[3], P.C[3], P.rw, vRx, vRy, s[6], c.cd[3], c.sd[3], c.muFz[3],
                        fx[3], fy[3], fxt[3], fyt[3], sl[3]);

    // :376-385
    const T Udot = P.inv_m * (fx[0] + fx[1] + fx[2] + fx[3]) + V * wz;
    const T Vdot = P.inv_m * (fy[0] + fy[1] + fy[2] + fy[3]) - U * wz;
    k[0] = Udot;
    k[1] = Vdot;
    k[2] = P.inv_Izz * (P.a * (fy[0] + fy[1]) - P.b * (fy[2] + fy[3])
                        + P.half_T * (fx[1] - fx[0] + fx[3] - fx[2]));
    // quirk Q2: front wheels see the tire-frame force, rear wheels the chassis-frame one
    k[3] = (c.tq[0] - P.rw * fxt[0]) * P.inv_Jw;
    k[4] = (c.tq[1] - P.rw * fxt[1]) * P.inv_Jw;
    k[5] = (c.tq[2] - P.rw * fx[2]) * P.inv_Jw;
    k[6] = (c.tq[3] - P.rw * fx[3]) * P.inv_Jw;
    k[7] = wz;
    k[8] = U * cy - V * sy;
    k[9] = U * sy + V * cy;
    axc = Udot - V * wz;  // :413
    ayc = Vdot + U * wz;  // :414
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

// Classic RK4 with frozen inputs, vehicle_model.py:427-445.  Updates s[10] in
// place and replaces (ax, ay) by the 1-2-2-1 averages of axc, ayc (:442-443).
// DIAG: also returns state_dot (:440) and the averaged outputs (:441).
template <typename T, bool REAR, bool DIAG>
__device__ __forceinline__ void rk4_step(const DevParams<T> &P, T s[10], T &ax, T &ay, const T delta[4],
                                         const T tq[4], const T mu[4], T h, T *state_dot,
                                         Outputs18<T> *outputs)
{
    StepInv<T> c;
    make_step_inv<T, REAR>(P, delta, tq, mu, ax, ay, c);
    const T hh = T(0.5) * h;
    T k[10], acc[10], st[10], a1, a2, asx, asy;
    Outputs18<T> o, osum;

    // sin / cos of the stage yaw: one full evaluation per step (quirk Q7: yaw is
    // never wrapped), then the stage increments d = yaw_stage - yaw are rotated
    // in with a short series; a large increment falls back to the full evaluation.
    using M = Math<T>;
    T sy0, cy0, sy, cy, sdl, cdl;
    M::sincos(s[7], &sy0, &cy0);
#define VDYN_STAGE_SINCOS(dexpr)                                        \
    do {                                                                \
        const T d_ = (dexpr);                                           \
        if (M::small_sincos(d_, &sdl, &cdl)) {                          \
            sy = sy0 * cdl + cy0 * sdl;                                 \
            cy = cy0 * cdl - sy0 * sdl;                                 \
        } else {                                                        \
            M::sincos(st[7], &sy, &cy);                                 \
        }                                                               \
    } while (0)

    planar_deriv<T, REAR, DIAG>(P, c, s, sy0, cy0, k, a1, a2, &o);  // K1 (:429)
    asx = a1; asy = a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] = k[i]; st[i] = s[i] + hh * k[i]; }
    if (DIAG) osum = o;

    VDYN_STAGE_SINCOS(hh * k[7]);
    planar_deriv<T, REAR, DIAG>(P, c, st, sy, cy, k, a1, a2, &o);   // K2 (:431)
    asx += T(2) * a1; asy += T(2) * a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] += T(2) * k[i]; st[i] = s[i] + hh * k[i]; }
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 18; ++i) osum.v[i] += T(2) * o.v[i];
    }

    VDYN_STAGE_SINCOS(hh * k[7]);
    planar_deriv<T, REAR, DIAG>(P, c, st, sy, cy, k, a1, a2, &o);   // K3 (:433)
    asx += T(2) * a1; asy += T(2) * a2;
#pragma unroll
    for (int i = 0; i < 10; ++i) { acc[i] += T(2) * k[i]; st[i] = s[i] + h * k[i]; }
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 18; ++i) osum.v[i] += T(2) * o.v[i];
    }

    VDYN_STAGE_SINCOS(h * k[7]);
#undef VDYN_STAGE_SINCOS
    planar_deriv<T, REAR, DIAG>(P, c, st, sy, cy, k, a1, a2, &o);   // K4 (:435)
    asx += a1; asy += a2;
    const T h6 = h * T(1.0 / 6.0), sixth = T(1.0 / 6.0);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        acc[i] += k[i];
        s[i] += h6 * acc[i];                                       // :438
    }
    ax = asx * sixth;                                              // :442
    ay = asy * sixth;                                              // :443
    if (DIAG) {
#pragma unroll
        for (int i = 0; i < 10; ++i) state_dot[i] = acc[i] * sixth;        // :440
#pragma unroll
        for (int i = 0; i < 18; ++i) outputs->v[i] = (osum.v[i] + o.v[i]) * sixth;  // :441
    }
}

}  // namespace vdyn
