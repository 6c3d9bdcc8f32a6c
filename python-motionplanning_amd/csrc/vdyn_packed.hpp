// vdyn_packed.hpp -- the fp32 FAST step on packed (2 x fp32) VALU instructions.
//
// Why: at one wave per SIMD -- what 65536 rollouts give this chip -- throughput is bound by
// how often ONE wave can issue, not by the ALUs: a lone wave issues one VALU instruction per
// ~4.6 cycles whether it is v_fma_f32 or v_pk_fma_f32 (tools/ubench/dep_latency.hip,
// profiles/r01_ubench_dep_latency.txt), so a packed instruction does two lanes' worth of work in
// one issue slot.  (With 2+ waves per SIMD a packed op costs 1.7 scalar ops: still a gain.)
// The model pairs up naturally: (FL, FR) and (RL, RR) run the same tire computation, (sin, cos)
// share a Horner chain, (U_dot, V_dot), (x_dot, y_dot), (axc, ayc) are 2-D rotations / cross
// products, and the ten states advance as five pairs.
//
// What cannot be packed stays scalar on the halves of a pair: v_rcp / v_rsq, min / max, |x|.
// The tire chain of the FAST step has no compare, select or argument reduction at all (pacejka_g2x2).
// Polynomial coefficients live in VGPR pairs for the whole kernel (VOP3P takes no literal constants
// on gfx9); PkConsts::init and PkParams::init pin them there.
//
// Semantics are those of vdyn_device.hpp (same formulas, same quirk handling); only the order
// of the four-tire sums differs ((FL+RL)+(FR+RR) instead of ((FL+FR)+RL)+RR).
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_device.hpp"

namespace vdyn {

typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f2 splat(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// Packed multiplies with a per-half sign, which VOP3P encodes for free (neg_lo / neg_hi) but the
// compiler only forms for whole-vector negation.  Half swizzles (op_sel) it does form by itself.
// Operands must not come straight from v_rcp / v_rsq: the wait state a transcendental result
// needs is the compiler's to insert, and it does not look inside inline asm.
__device__ __forceinline__ f2 pk_cross(f2 a, f2 b)        // (a.y b.x, -a.x b.x)
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f2 pk_hi_conj(f2 a, f2 b)      // (-a.y b.x, a.y b.y)
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Coefficients and other constants as (c, c) pairs, made opaque to the optimiser so that they are
// materialised once and stay in VGPRs instead of being rebuilt at every use.
struct PkConsts {
    f2 at[8];            // atan(t) = t Q(t^2) on [0, 1]: q7 .. q0 (tools/fit_polys.py, degree 7: 2.3e-7 relative)
    f2 sn[4];            // sin on [-pi/2, pi/2]: the four coefficients of sin_mid / sin_0_pi
    f2 pi_hi, pi_lo, inv_pi;
    f2 neg2, tiny;       // -2, 1e-30 (floor of s^2, quirk Q5)
    f2 side;             // (-1, +1): left / right wheel (quirk Q8)
    f2 sck[3];           // (sin, cos) kernel coefficients on |r| <= pi/4 (sincos_kernel), one pair per degree
    f2 rot_a, rot_b, rot_c;   // (-1/6, 1/24), (1, -1/2), (0, 1): (sin d, cos d) of a stage's small yaw increment
    f2 scp[4];           // (sin, cos) on |r| <= pi/2 in one chain: (sin, cos) coefficient pairs (sincos_mid2)

    __device__ __forceinline__ void pin(f2 &v, float c)
    {
        v = splat(c);
        asm volatile("" : "+v"(v));
    }
    // FIT: the tire chain is pacejka_g2x2 -- the atan / cosine coefficients are not needed (and not pinned)
    template <bool FIT = false>
    __device__ __forceinline__ void init()
    {
        const float a[8] = {-4.729942884e-03f, 2.439327165e-02f, -5.969851837e-02f, 9.930104017e-02f,
                            -1.402552277e-01f, 1.997082233e-01f, -3.333206475e-01f, 9.999998808e-01f};
        const float s[4] = {2.607052693e-06f, -1.981028618e-04f, 8.333077654e-03f, -1.666665971e-01f};
        if (!FIT) {
#pragma unroll
            for (int i = 0; i < 8; ++i) pin(at[i], a[i]);
        }
        if (!FIT) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pin(sn[i], s[i]);
            pin(pi_hi, 3.1415927410125732421875f);
            pin(pi_lo, -8.74227800037248566e-08f);
            pin(inv_pi, 0.318309886183790671538f);
            pin(tiny, 1e-30f);
        }
        pin(neg2, -2.0f);
        side = f2{-1.0f, 1.0f};
        asm volatile("" : "+v"(side));
        const float sks[3] = {-1.951163867e-04f, 8.332134224e-03f, -1.666665375e-01f};
        const float cks[3] = {2.443367339e-05f, -1.388732577e-03f, 4.166664556e-02f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sck[i] = f2{sks[i], cks[i]};
            asm volatile("" : "+v"(sck[i]));
        }
        // cos(z) on |z| <= pi/2 as a degree-4 polynomial in z^2 (tools/fit_polys.py: fit error 5.1e-8,
        // fp32 Horner within 1.8e-7 absolute)
        const float cc[5] = {2.312937249e-05f, -1.385257230e-03f, 4.166342318e-02f, -4.999989867e-01f, 9.999999404e-01f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            scp[i] = f2{s[i], cc[i]};
            asm volatile("" : "+v"(scp[i]));
        }
        rot_a = f2{-1.0f / 6.0f, 1.0f / 24.0f};
        rot_b = f2{1.0f, -0.5f};
        rot_c = f2{0.0f, 1.0f};
        asm volatile("" : "+v"(rot_a), "+v"(rot_b), "+v"(rot_c));
    }
};

// Vehicle constants as pairs (front axle pair F = (FL, FR), rear pair R = (RL, RR)).
struct PkParams {
    f2 BF, BR, invBF, invBR, CF, CR, rw, inv_Jw;
    f2 ab_F, ab_R;                                       // (+a, +a) / Izz, (-b, -b) / Izz: lever arms of the pairs
    f2 hT_Izz;                                           // (-T/2, +T/2) / Izz: moment arms of the (left, right) longitudinal forces
    f2 a_negb, hT_side, inv_m;                           // (a, -b); (-T/2, +T/2); (1/m, 1/m)
    f2 neg_rw_Jw;                                        // -rw / Jw
    f2 Fz0F, Fz0R, dfxF, dfyF, dfxR, dfyR;               // static loads and load-transfer coefficients per pair (:255-258)
    f2 fwF[kTireFitDeg + 1], fwR[kTireFitDeg + 1];       // W_C(c) per wheel (TireFit), highest degree first (FIT only)
    // UNIFORM: P is the same for every lane (a by-value kernel argument, in SGPRs); otherwise it was
    // read per lane (heterogeneous fleet, VGPRs).
    template <bool UNIFORM, bool FIT = false>
    __device__ __forceinline__ void init(const DevParams<float> &P)
    {
        if (FIT) {
#pragma unroll
            for (int i = 0; i <= kTireFitDeg; ++i) {
                fwF[i] = f2{P.W[i][0], P.W[i][1]};
                fwR[i] = f2{P.W[i][2], P.W[i][3]};
                asm volatile("" : "+v"(fwF[i]), "+v"(fwR[i]));   // nine pairs per axle, read at every stage: VGPRs
            }
        }
        BF = f2{P.B[0], P.B[1]}; BR = f2{P.B[2], P.B[3]};
        invBF = f2{P.invB[0], P.invB[1]}; invBR = f2{P.invB[2], P.invB[3]};
        CF = f2{P.C[0], P.C[1]}; CR = f2{P.C[2], P.C[3]};
        // Scalars first, each behind a zero-instruction barrier: left visible as neighbouring struct
        // fields, pairs of them are fetched with one 8-byte read, and because such reads overlap
        // (inv_Jw|a, a|b) the compiler then parks that part of the by-value struct in scratch.
        float a = P.a, b = P.b, iJw = P.inv_Jw, dxl = P.DfzxL, dxr = P.DfzxR, dyf = P.DfzyF, dyr = P.DfzyR,
              r_w = P.rw, f0f = P.Fz0F, f0r = P.Fz0R, hT = P.half_T, im = P.inv_m, iIz = P.inv_Izz;
        if (UNIFORM)   // stay in SGPRs: a scalar source operand costs the packed op no VGPR read port
            asm("" : "+s"(a), "+s"(b), "+s"(iJw), "+s"(dxl), "+s"(dxr), "+s"(dyf), "+s"(dyr), "+s"(r_w), "+s"(f0f),
                "+s"(f0r), "+s"(hT), "+s"(im), "+s"(iIz));
        else
            asm("" : "+v"(a), "+v"(b), "+v"(iJw), "+v"(dxl), "+v"(dxr), "+v"(dyf), "+v"(dyr), "+v"(r_w), "+v"(f0f),
                "+v"(f0r), "+v"(hT), "+v"(im), "+v"(iIz));
        rw = splat(r_w); inv_Jw = splat(iJw);
        ab_F = splat(a * iIz); ab_R = splat(-b * iIz);
        hT_Izz = f2{-hT * iIz, hT * iIz};
        a_negb = f2{a, -b};
        hT_side = f2{-hT, hT};
        inv_m = splat(im);
        neg_rw_Jw = splat(-r_w * iJw);
        Fz0F = splat(f0f); Fz0R = splat(f0r);
        dfxF = f2{-dxl, -dxr}; dfyF = f2{-dyf, dyf};
        dfxR = f2{dxl, dxr}; dfyR = f2{-dyr, dyr};
        // used once per step: keep them in VGPRs, the SGPR file is already full of DevParams
        asm volatile("" : "+v"(Fz0F), "+v"(Fz0R), "+v"(dfxF), "+v"(dfyF), "+v"(dfxR), "+v"(dfyR), "+v"(neg_rw_Jw));
    }
};

// The front pair (FL, FR) and the rear pair (RL, RR) are evaluated in lockstep, statement by
// statement: a packed instruction may not be followed immediately by one that reads its result
// (the compiler pads with s_nop, which costs a lone wave a whole issue slot), and the other
// pair's independent instruction is exactly the filler that makes the padding unnecessary.
#define VDYN_BOTH(q) _Pragma("unroll") for (int q = 0; q < 2; ++q)

// sin(C atan(x)) for both pairs, any C and any sign of x (handles without a validated fit); inv_x = 1/x per half.
__device__ __forceinline__ void sin_c_atan2x2(const PkConsts &K, const f2 C[2], const f2 x[2], const f2 inv_x[2],
                                              f2 out[2])
{
    // atan_rcp on all four halves: selects scalar, Horner chain packed
    bool b0[2], b1[2];
    f2 t[2], u[2], p[2], th[2], y[2], r[2], w[2], ps[2], kk[2];
    VDYN_BOTH(q) { b0[q] = ::fabsf(x[q].x) > 1.0f; b1[q] = ::fabsf(x[q].y) > 1.0f; }
    VDYN_BOTH(q) t[q] = f2{b0[q] ? inv_x[q].x : x[q].x, b1[q] ? inv_x[q].y : x[q].y};
    VDYN_BOTH(q) u[q] = t[q] * t[q];
    VDYN_BOTH(q) p[q] = K.at[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) VDYN_BOTH(q) p[q] = fma2(p[q], u[q], K.at[i]);
    VDYN_BOTH(q) p[q] = p[q] * t[q];
    {
        const float pio2 = 1.57079637050628662109375f;
        VDYN_BOTH(q) th[q] = f2{b0[q] ? (::copysignf(pio2, x[q].x) - p[q].x) : p[q].x,
                               b1[q] ? (::copysignf(pio2, x[q].y) - p[q].y) : p[q].y};
    }
    VDYN_BOTH(q) y[q] = C[q] * th[q];
    {
        VDYN_BOTH(q) kk[q] = y[q] * K.inv_pi;
        VDYN_BOTH(q) kk[q] = f2{__builtin_rintf(kk[q].x), __builtin_rintf(kk[q].y)};
        VDYN_BOTH(q) r[q] = fma2(-kk[q], K.pi_hi, y[q]);
        VDYN_BOTH(q) r[q] = fma2(-kk[q], K.pi_lo, r[q]);
    }
    VDYN_BOTH(q) w[q] = r[q] * r[q];
    VDYN_BOTH(q) ps[q] = fma2(K.sn[0], w[q], K.sn[1]);
    VDYN_BOTH(q) ps[q] = fma2(ps[q], w[q], K.sn[2]);
    VDYN_BOTH(q) ps[q] = fma2(ps[q], w[q], K.sn[3]);
    VDYN_BOTH(q) w[q] = r[q] * w[q];
    VDYN_BOTH(q) out[q] = fma2(w[q], ps[q], r[q]);
    {
        VDYN_BOTH(q) {
            const unsigned f0 = ((unsigned)(int)kk[q].x) << 31, f1 = ((unsigned)(int)kk[q].y) << 31;
            out[q] = f2{__uint_as_float(__float_as_uint(out[q].x) ^ f0), __uint_as_float(__float_as_uint(out[q].y) ^ f1)};
        }
    }
}

// G(x) = sin(C atan x) / x for both pairs from 1 + x^2 (x = B s, any sign of B: G is even in x):
// c = rsq(1 + x^2), G = c W_C(c) with the handle's own polynomial (TireFit in vdyn_device.hpp).  Eight packed
// fmas and one multiply per pair for every slip -- the chain it replaces (x and 1/x from rsq(s^2), min, indicator,
// degree-7 atan, phase, degree-4 cosine) took 21.  x = 0 needs no special case (quirk Q5: G(0) = C, and the
// reference's s == 0 branch returns what s -> 0 gives); x = inf gives c = 0, G = 0.  Returns c W_C(c) * scale.
__device__ __forceinline__ void pacejka_g2x2(const f2 *const W[2], const f2 q1[2], const f2 scale[2], f2 out[2])
{
    f2 c[2], g[2], cm[2];
    VDYN_BOTH(q) c[q] = f2{fm::rsq(q1[q].x), fm::rsq(q1[q].y)};
    VDYN_BOTH(q) g[q] = fma2(W[q][0], c[q], W[q][1]);
#pragma unroll
    for (int i = 2; i <= kTireFitDeg; ++i) VDYN_BOTH(q) g[q] = fma2(g[q], c[q], W[q][i]);
    VDYN_BOTH(q) cm[q] = c[q] * scale[q];                      // beside the Horner chain, not behind it
    VDYN_BOTH(q) out[q] = g[q] * cm[q];
}

// All four tires (vehicle_model.py:274-373, as tire_force in vdyn_device.hpp): index 0 = front
// pair (always steered), index 1 = rear pair (steered only with k = 12 controls).
// CS (B >= 0 and the handle's fit validated): the slips are carried pre-multiplied by B and
// s_x mu / s = (B s_x) G(B s) comes from pacejka_g2x2 -- no 1/s, no s at all.
// s[] (DIAG only) is the combined slip itself.
template <bool REAR_STEERED, bool CS, bool DIAG>
__device__ __forceinline__ void tire_force2x2(const PkConsts &K, const PkParams &Q, f2 vxc, const f2 vyc[2],
                                              const f2 w[2], const f2 cd[2], const f2 sd[2], const f2 muFz[2], f2 fx[2],
                                              f2 fy[2], f2 fxt[2], f2 fyt[2], f2 s[2])
{
    const f2 B[2] = {Q.BF, Q.BR}, invB[2] = {Q.invBF, Q.invBR}, C[2] = {Q.CF, Q.CR};
    const f2 rw = Q.rw;
    f2 vx[2], vy[2], rvx[2], sx[2], sy[2], s2[2], rs[2], xs[2], ix[2], g[2], tmp[2];
    tmp[0] = vyc[0] * sd[0];
    vx[0] = fma2(vxc, cd[0], tmp[0]);
    tmp[0] = vxc * sd[0];
    vy[0] = fma2(vyc[0], cd[0], -tmp[0]);
    if (REAR_STEERED) {
        tmp[1] = vyc[1] * sd[1];
        vx[1] = fma2(vxc, cd[1], tmp[1]);
        tmp[1] = vxc * sd[1];
        vy[1] = fma2(vyc[1], cd[1], -tmp[1]);
    } else {
        vx[1] = vxc;
        vy[1] = vyc[1];
    }
    VDYN_BOTH(q) rvx[q] = f2{fm::rcp(vx[q].x), fm::rcp(vx[q].y)};
    if (CS) VDYN_BOTH(q) rvx[q] = rvx[q] * B[q];                                                   // B / vx
    VDYN_BOTH(q) sx[q] = fma2(rw, w[q], -vx[q]);
    VDYN_BOTH(q) sx[q] = sx[q] * rvx[q];
    VDYN_BOTH(q) sy[q] = f2{-vy[q].x * ::fabsf(rvx[q].x), -vy[q].y * ::fabsf(rvx[q].y)};          // quirk Q4
    if (CS) {
        const f2 *const Wq[2] = {Q.fwF, Q.fwR};
        const f2 one = splat(1.0f);
        VDYN_BOTH(q) s2[q] = fma2(sy[q], sy[q], one);
        VDYN_BOTH(q) s2[q] = fma2(sx[q], sx[q], s2[q]);                                           // 1 + (B s)^2
        pacejka_g2x2(Wq, s2, muFz, g);
        if (DIAG) {
            // the combined slip itself, for the log only: |x| / B = x^2 rsq(x^2) / B, exactly 0 at x = 0 (quirk Q5)
            const f2 tiny = splat(1e-30f);
            VDYN_BOTH(q) xs[q] = fma2(sx[q], sx[q], sy[q] * sy[q]);
            VDYN_BOTH(q) rs[q] = f2{fm::rsq(::fmaxf(xs[q].x, tiny.x)), fm::rsq(::fmaxf(xs[q].y, tiny.y))};
            VDYN_BOTH(q) s[q] = xs[q] * rs[q] * invB[q];
        }
    } else {
        // quirk Q5: s == 0 takes the fallback branch in the reference, whose value is what the regular
        // formula yields for s -> 0; s^2 + 1e-30 keeps rsq finite there and is s^2 to the last bit
        // wherever s^2 >= 1e-22 (below that the force is linear in slip and does not see s at all)
        VDYN_BOTH(q) s2[q] = fma2(sy[q], sy[q], K.tiny);
        VDYN_BOTH(q) s2[q] = fma2(sx[q], sx[q], s2[q]);
        VDYN_BOTH(q) rs[q] = f2{fm::rsq(s2[q].x), fm::rsq(s2[q].y)};
        VDYN_BOTH(q) s[q] = s2[q] * rs[q];
        VDYN_BOTH(q) xs[q] = B[q] * s[q];
        VDYN_BOTH(q) ix[q] = rs[q] * invB[q];
        sin_c_atan2x2(K, C, xs, ix, g);
        VDYN_BOTH(q) g[q] = g[q] * rs[q];
        VDYN_BOTH(q) g[q] = g[q] * muFz[q];
    }
    VDYN_BOTH(q) fxt[q] = sx[q] * g[q];
    VDYN_BOTH(q) fyt[q] = sy[q] * g[q];
    tmp[0] = fyt[0] * sd[0];
    fx[0] = fma2(fxt[0], cd[0], -tmp[0]);
    tmp[0] = fyt[0] * cd[0];
    fy[0] = fma2(fxt[0], sd[0], tmp[0]);
    if (REAR_STEERED) {
        tmp[1] = fyt[1] * sd[1];
        fx[1] = fma2(fxt[1], cd[1], -tmp[1]);
        tmp[1] = fyt[1] * cd[1];
        fy[1] = fma2(fxt[1], sd[1], tmp[1]);
    } else {
        fx[1] = fxt[1];
        fy[1] = fyt[1];
    }
}

// The ten states as five pairs.
// (sin r, cos r) for |r| <= pi/4: vdyn_fastmath.hpp's sincos_kernel with both Horner chains in
// one packed chain.
__device__ __forceinline__ f2 sincos_kernel2(const PkConsts &K, float r)
{
    const float u = r * r;
    const f2 u2 = f2{u, u};
    f2 p = fma2(K.sck[0], u2, K.sck[1]);
    p = fma2(p, u2, K.sck[2]);
    const f2 a = f2{r, u} * u2;                                 // (r u, u u)
    return fma2(a, p, f2{r, ::fmaf(-0.5f, u, 1.0f)});
}

// (sin x, cos x) for |x| <= fm::kSincosMidLimit (yaw is never wrapped, quirk Q7): x = k pi + r with
// |r| <= pi/2 (two-term Cody-Waite: the neglected tail of pi is 3e-15 k), sin r as the odd and cos r as
// the even polynomial in ONE packed Horner chain, both signs flipped for odd k.  No quadrant swap.
// sin: 1.1e-7 relative; cos: 1.8e-7 absolute (it multiplies velocities of tens of m/s: 5e-6 m/s).
__device__ __forceinline__ f2 sincos_mid2(const PkConsts &K, float x, bool &ok)
{
    const float k = __builtin_rintf(x * 0.318309886183790671538f);
    float r = ::fmaf(-k, 3.1415927410125732421875f, x);
    r = ::fmaf(-k, -8.74227800037248566e-08f, r);
    const float u = r * r;
    const f2 u2 = f2{u, u};
    f2 p = fma2(K.scp[0], u2, K.scp[1]);
    p = fma2(p, u2, K.scp[2]);
    p = fma2(p, u2, K.scp[3]);                                  // (S(u), c0 + c1 u + c2 u^2 + c3 u^3 -> times u + c4 below)
    // the two last Horner steps as scalar fmas (the pairs (r u, u) and (r, c4) would each cost a register move)
    const float sn = ::fmaf(r * u, p.x, r);                     // r + r u S
    const float cs = ::fmaf(u, p.y, 9.999999404e-01f);          // c4 + u (...)
    const unsigned flip = ((unsigned)(int)k) << 31;
    ok = ok && (::fabsf(x) <= fm::kSincosMidLimit);
    return f2{__uint_as_float(__float_as_uint(sn) ^ flip), __uint_as_float(__float_as_uint(cs) ^ flip)};
}

// (sin, cos) of a steering angle: no reduction inside |delta| <= pi/4 (every physical steering
// range; max_steer is 30 deg in drive.py:51); beyond it the lane takes the SAFE step.
__device__ __forceinline__ f2 sincos_steer2(const PkConsts &K, float d, bool &ok)
{
    ok = ok && (::fabsf(d) <= fm::kSincosKernelLimit);
    return sincos_kernel2(K, d);
}

// (sin d, cos d) of a stage's yaw increment d = (h/2 or h) wz, |d| <= kStageYawLimit: the Taylor
// forms d (1 - d^2/6) and 1 - d^2/2 + d^4/24 (relative error 8e-9 and 1.3e-12 at the limit) in
// two packed operations.  Larger increments (wz > 31 rad/s at dt = 1e-3) send the lane to SAFE.
constexpr float kStageYawLimit = 0.03125f;
__device__ __forceinline__ f2 stage_rot2(const PkConsts &K, float d)
{
    const float u = d * d;
    const f2 a = fma2(f2{u, u}, K.rot_a, K.rot_b);             // (1 - u/6, -1/2 + u/24)
    return fma2(f2{d, u}, a, K.rot_c);                         // (d (1 - u/6), 1 + u (-1/2 + u/24))
}

struct State5 {
    f2 uv;    // U, V
    f2 wy;    // wz, yaw
    f2 wf;    // wFL, wFR
    f2 wr;    // wRL, wRR
    f2 xy;    // x, y
};

struct StepInv2 {
    f2 cdF, sdF, cdR, sdR;   // cos / sin of the steering angles per pair
    f2 muFzF, muFzR;         // mu_max * Fz
    f2 tqF, tqR;             // torque / Jw
};

// Diagnostics of one derivative evaluation as pairs (vehicle_model.py:420-423): chassis-frame
// forces and combined slips of the front / rear pair, tire-frame forces of the front pair.
struct Diag2 {
    f2 fx[2], fy[2], s[2], fxt, fyt;
};

// vehicle_model.py:220-425 on pairs.  sc = (sin, cos) of the stage yaw MINUS the yaw the step
// started from: k.xy is the global-frame velocity (:384-385) expressed in the frame of the initial
// yaw, and rk4_step2 rotates the 1-2-2-1 sum into the global frame once (a rotation is linear, so
// R(yaw0) sum_j w_j R(d_j) u_j == sum_j w_j R(yaw0 + d_j) u_j).  FIRST: the first stage, d = 0.
template <bool K2, bool CS, bool DIAG = false, bool FIRST = false>
__device__ __forceinline__ void planar_deriv2(const DevParams<float> &P, const PkParams &Q, const PkConsts &K,
                                              const StepInv2 &c, const State5 &s, f2 sc, State5 &k, f2 &sums_out,
                                              Diag2 *dg = nullptr)
{
    const float U = s.uv.x, V = s.uv.y, wz = s.wy.x;
    const f2 U2 = f2{U, U}, V2 = f2{V, V}, wz2 = f2{wz, wz};
    const f2 vxc = fma2(Q.hT_side, wz2, U2);                  // :261-271 (left, right), both axles (quirk Q8)
    const f2 vyb = fma2(Q.a_negb, wz2, V2);                   // (V + a wz, V - b wz): front, rear
    const f2 vyq[2] = {f2{vyb.x, vyb.x}, f2{vyb.y, vyb.y}}, wq[2] = {s.wf, s.wr};
    const f2 cdq[2] = {c.cdF, c.cdR}, sdq[2] = {c.sdF, c.sdR}, mfq[2] = {c.muFzF, c.muFzR};
    f2 fxq[2], fyq[2], fxtq[2], fytq[2], slq[2];
    tire_force2x2<!K2, CS, DIAG>(K, Q, vxc, vyq, wq, cdq, sdq, mfq, fxq, fyq, fxtq, fytq, slq);
    if (DIAG) {
        dg->fx[0] = fxq[0]; dg->fx[1] = fxq[1];
        dg->fy[0] = fyq[0]; dg->fy[1] = fyq[1];
        dg->s[0] = slq[0]; dg->s[1] = slq[1];
        dg->fxt = fxtq[0]; dg->fyt = fytq[0];
    }
    const f2 fxF = fxq[0], fyF = fyq[0], fxtF = fxtq[0], fxR = fxq[1], fyR = fyq[1];
    // :376-378
    const f2 sfx = fxF + fxR, sfy = fyF + fyR;                 // (left sums, right sums)
    float sumx = sfx.x + sfx.y, sumy = sfy.x + sfy.y;
    asm("" : "+v"(sumx));                                      // keep the two adds scalar: packing them costs 3 v_mov
    const f2 sums = f2{sumx, sumy};                            // (sum Fx, sum Fy)
    const f2 cross = pk_cross(s.uv, s.wy);                     // (V wz, -U wz)
    k.uv = fma2(Q.inv_m, sums, cross);                         // (U_dot, V_dot)
    // :413-414: axc = U_dot - V wz, ayc = V_dot + U wz are the force sums over m; only their 1-2-2-1
    // average is ever used (:442-443), so the caller accumulates the sums and scales once
    sums_out = sums;
    f2 my = fma2(fyF, Q.ab_F, fyR * Q.ab_R);                   // (a fy_front - b fy_rear) / Izz, per side
    my = fma2(sfx, Q.hT_Izz, my);                              // -+ T/2 fx / Izz, per side (:378, quirk Q8)
    const float wzdot = my.x + my.y;
    k.wy = f2{wzdot, wz};
    k.wf = fma2(Q.neg_rw_Jw, fxtF, c.tqF);                     // :379-382, quirk Q2: tire-frame force in front,
    k.wr = fma2(Q.neg_rw_Jw, fxR, c.tqR);                      //           chassis-frame force at the rear
    if (FIRST) k.xy = s.uv;
    else k.xy = fma2(U2, f2{sc.y, sc.x}, pk_hi_conj(s.uv, sc));   // :384-385 (U cd - V sd, U sd + V cd)
}

// FAST RK4 step (vehicle_model.py:427-445) on pairs; `ok` as in rk4_step.
// Returns the 1-2-2-1 sum of the stage derivatives in `acc` (x, y already in the global frame): the
// caller finishes with s += h/6 acc (:438) -- in place when every lane of the wave stayed in range,
// see StepEngine<float>::advance.
// DIAG: also state_dot (:440) and the 1-2-2-1 averaged outputs (:441).
// PRE: (sin, cos) of the front steering angle arrive precomputed (LDS-shared control tables hold
// them per entry; computed by the same sincos_kernel2, so the result is bit for bit the same).
// PRE = 2: delta[0] holds tan(delta) and (sin, cos) are exact for any angle (no range check).
template <bool K2, bool CS, bool DIAG = false, int PRE = 0>
__device__ __forceinline__ bool rk4_step2(const DevParams<float> &P, const PkParams &Q, const PkConsts &K,
                                          const State5 &s, f2 axy, const float delta[4], const float tq[4],
                                          const float mu[4], float h, State5 &acc, f2 &axy_n,
                                          State5 *state_dot = nullptr, Diag2 *outputs = nullptr, f2 *FzF_out = nullptr,
                                          f2 *FzR_out = nullptr, f2 steer_sc = f2{0.0f, 1.0f})
{
    bool ok = true;
    StepInv2 c;
    {
        f2 d0;
        if (PRE != 0) {
            d0 = steer_sc;
            if (PRE == 1) ok = ok && (::fabsf(delta[0]) <= fm::kSincosKernelLimit);
        } else {
            d0 = sincos_steer2(K, delta[0], ok);
        }
        if (K2) {
            c.sdF = f2{d0.x, d0.x}; c.cdF = f2{d0.y, d0.y};
            c.sdR = splat(0.0f); c.cdR = splat(1.0f);
        } else {
            const f2 d1 = sincos_steer2(K, delta[1], ok), d2 = sincos_steer2(K, delta[2], ok),
                     d3 = sincos_steer2(K, delta[3], ok);
            c.sdF = f2{d0.x, d1.x}; c.cdF = f2{d0.y, d1.y};
            c.sdR = f2{d2.x, d3.x}; c.cdR = f2{d2.y, d3.y};
        }
        // :255-258 (quirk Q3), then mu_max * Fz (quirk Q1)
        const f2 ax2 = f2{axy.x, axy.x}, ay2 = f2{axy.y, axy.y};
        const f2 FzF = fma2(Q.dfyF, ay2, fma2(Q.dfxF, ax2, Q.Fz0F));
        const f2 FzR = fma2(Q.dfyR, ay2, fma2(Q.dfxR, ax2, Q.Fz0R));
        c.muFzF = f2{mu[0], mu[1]} * FzF;
        c.muFzR = f2{mu[2], mu[3]} * FzR;
        if (DIAG) { *FzF_out = FzF; *FzR_out = FzR; }
        c.tqF = f2{tq[0], tq[1]} * Q.inv_Jw;
        c.tqR = f2{tq[2], tq[3]} * Q.inv_Jw;
    }
    const float hh = 0.5f * h;
    const f2 hh2 = splat(hh), h2 = splat(h), two = splat(2.0f);
    f2 a, as2;
    const f2 sc0 = sincos_mid2(K, s.wy.y, ok);
    f2 sc;
    State5 k, st;
    float d2, d3, d4;

    // x, y and yaw feed no derivative (:376-385 read U, V, wz and the wheel speeds only), so the
    // stages carry the four dynamic pairs; acc.xy accumulates in the frame of the initial yaw.
#define VDYN_S4_EACH(OP) OP(uv) OP(wy) OP(wf) OP(wr)
#define VDYN_S5_EACH(OP) VDYN_S4_EACH(OP) OP(xy)
    Diag2 dg, dsum;
#define VDYN_DG_EACH(OP) OP(fx[0]) OP(fx[1]) OP(fy[0]) OP(fy[1]) OP(s[0]) OP(s[1]) OP(fxt) OP(fyt)
    planar_deriv2<K2, CS, DIAG, true>(P, Q, K, c, s, sc0, k, a, &dg);       // K1
    as2 = a;
    if (DIAG) dsum = dg;
#define VDYN_S5_1(f) acc.f = k.f; st.f = fma2(hh2, k.f, s.f);
    VDYN_S4_EACH(VDYN_S5_1)
    acc.xy = k.xy;
    d2 = hh * k.wy.y;
    sc = stage_rot2(K, d2);
    planar_deriv2<K2, CS, DIAG>(P, Q, K, c, st, sc, k, a, &dg);             // K2
    as2 = fma2(two, a, as2);
#define VDYN_DG_2(f) dsum.f = fma2(two, dg.f, dsum.f);
    if (DIAG) { VDYN_DG_EACH(VDYN_DG_2) }
#define VDYN_S5_2(f) acc.f = fma2(two, k.f, acc.f); st.f = fma2(hh2, k.f, s.f);
    VDYN_S4_EACH(VDYN_S5_2)
    acc.xy = fma2(two, k.xy, acc.xy);
    d3 = hh * k.wy.y;
    sc = stage_rot2(K, d3);
    planar_deriv2<K2, CS, DIAG>(P, Q, K, c, st, sc, k, a, &dg);             // K3
    as2 = fma2(two, a, as2);
    if (DIAG) { VDYN_DG_EACH(VDYN_DG_2) }
#define VDYN_S5_3(f) acc.f = fma2(two, k.f, acc.f); st.f = fma2(h2, k.f, s.f);
    VDYN_S4_EACH(VDYN_S5_3)
    acc.xy = fma2(two, k.xy, acc.xy);
    d4 = h * k.wy.y;
    sc = stage_rot2(K, d4);
    planar_deriv2<K2, CS, DIAG>(P, Q, K, c, st, sc, k, a, &dg);             // K4
    as2 = as2 + a;
    // one range test for the three stage increments
    ok = ok && (::fmaxf(::fmaxf(::fabsf(d2), ::fabsf(d3)), ::fabsf(d4)) <= kStageYawLimit);
    const float sixth = 1.0f / 6.0f;
#define VDYN_S5_4(f) acc.f = acc.f + k.f;
    VDYN_S4_EACH(VDYN_S5_4)
    acc.xy = acc.xy + k.xy;
    acc.xy = fma2(f2{acc.xy.x, acc.xy.x}, f2{sc0.y, sc0.x}, pk_hi_conj(acc.xy, sc0));   // into the global frame
    if (DIAG) {
        const f2 sixth2 = splat(sixth);
#define VDYN_S5_5(f) state_dot->f = acc.f * sixth2;
        VDYN_S5_EACH(VDYN_S5_5)
#undef VDYN_S5_5
#define VDYN_DG_4(f) outputs->f = (dsum.f + dg.f) * sixth2;
        VDYN_DG_EACH(VDYN_DG_4)
#undef VDYN_DG_4
    }
#undef VDYN_DG_2
#undef VDYN_DG_EACH
#undef VDYN_S5_1
#undef VDYN_S5_2
#undef VDYN_S5_3
#undef VDYN_S5_4
#undef VDYN_S5_EACH
#undef VDYN_S4_EACH
    axy_n = as2 * (Q.inv_m * splat(sixth));                  // wave-uniform factor: hoisted out of the time loop
    return ok;
}

}  // namespace vdyn

namespace vdyn {

// What a kernel's time loop calls: per-kernel constants + one step.  fp32 takes the packed
// FAST step (SAFE scalar redo for lanes that left its validated range, exactly as rk4_advance);
// fp64 has no packed VALU form and goes through rk4_advance.
template <typename T>
struct StepEngine {
    // what a kernel carries from step to step: the 12 persistent scalars of a rollout
    struct State {
        T s[10], ax, ay;
        __device__ __forceinline__ T get(int i) const { return i < 10 ? s[i] : (i == 10 ? ax : ay); }
        __device__ __forceinline__ void set(int i, T v) { if (i < 10) s[i] = v; else if (i == 10) ax = v; else ay = v; }
    };
    // FITSRC = 2: the kernel staged the per-wheel fp64 fit table in LDS (planar_deriv)
    template <bool K2, bool CS, int PRE = 0, int FITSRC = 1, bool COMP = false>       // COMP: fp32 only
    __device__ __forceinline__ void advance_state(const DevParams<T> &P, State &X, const T delta[4], const T tq[4],
                                                  const T mu[4], T h, T sd0 = T(0), T cd0 = T(1)) const
    {
        rk4_advance<T, K2, false, CS, PRE, FITSRC>(P, X.s, X.ax, X.ay, delta, tq, mu, h, nullptr, nullptr, sd0, cd0);
    }
    template <bool UNIFORM = true, bool FIT = false>
    __device__ __forceinline__ void init(const DevParams<T> &) {}
    __device__ __forceinline__ void steer_sincos(T d, T &sd, T &cd) const
    {
        bool ok = true;
        Math<T, false>::sincos_steer(d, &sd, &cd, ok);
    }
    // constants PER LANE (heterogeneous fleet): P lives in the lane's registers and carries one set of fp64 fit
    // coefficients (column 0)
    template <bool K2, bool CS, int PRE = 0>
    __device__ __forceinline__ void advance(const DevParams<T> &P, T s[10], T &ax, T &ay, const T delta[4],
                                            const T tq[4], const T mu[4], T h, T sd0 = T(0), T cd0 = T(1)) const
    {
        rk4_advance<T, K2, false, CS, PRE, 1>(P, s, ax, ay, delta, tq, mu, h, nullptr, nullptr, sd0, cd0);
    }
    // the same step with state_dot [10] and the averaged outputs [18] (vehicle_model.py:440-441)
    template <bool K2, bool CS, int FITSRC = 1>
    __device__ __forceinline__ void advance_diag(const DevParams<T> &P, T s[10], T &ax, T &ay, const T delta[4],
                                                 const T tq[4], const T mu[4], T h, T sd[10], Outputs18<T> &o) const
    {
        rk4_advance<T, K2, true, CS, 0, FITSRC>(P, s, ax, ay, delta, tq, mu, h, sd, &o);
    }
};

template <>
struct StepEngine<float> {
    PkConsts K;
    PkParams Q;
    // FIT: the kernel's CS flag -- the step then takes the handle's fitted tire chain (pacejka_g2x2)
    template <bool UNIFORM = true, bool FIT = false>
    __device__ __forceinline__ void init(const DevParams<float> &P)
    {
        K.template init<FIT>();
        Q.template init<UNIFORM, FIT>(P);
    }
    // (sin, cos) of a steering angle exactly as the FAST step computes it (for control tables
    // that carry them per entry)
    __device__ __forceinline__ void steer_sincos(float d, float &sd, float &cd) const
    {
        const f2 r = sincos_kernel2(K, d);
        sd = r.x;
        cd = r.y;
    }
    // What a kernel carries from step to step: the ten states as five (even-aligned) register pairs
    // plus (ax_prev, ay_prev) -- the form the packed step reads and writes, so the time loop neither
    // unpacks nor re-packs anything.
    struct State {
        f2 uv, wy, wf, wr, xy, axy;
        f2 cuv, cwy, cwf, cwr, cxy;          // compensation terms of the five pairs (COMP kernels only; dead elsewhere)
        // rows of the [12][N] layout: U V wz wFL wFR wRL wRR yaw x y ax ay; rows 12..21 of the [22][N] layout
        // (VDYN_OPT_STATE_ROWS): the compensation terms of rows 0..9
        __device__ __forceinline__ float get(int i) const
        {
            switch (i) {
            case 0: return uv.x; case 1: return uv.y; case 2: return wy.x; case 3: return wf.x; case 4: return wf.y;
            case 5: return wr.x; case 6: return wr.y; case 7: return wy.y; case 8: return xy.x; case 9: return xy.y;
            case 10: return axy.x; case 11: return axy.y;
            case 12: return cuv.x; case 13: return cuv.y; case 14: return cwy.x; case 15: return cwf.x; case 16: return cwf.y;
            case 17: return cwr.x; case 18: return cwr.y; case 19: return cwy.y; case 20: return cxy.x; default: return cxy.y;
            }
        }
        __device__ __forceinline__ void set(int i, float v)
        {
            switch (i) {
            case 0: uv.x = v; break; case 1: uv.y = v; break; case 2: wy.x = v; break; case 3: wf.x = v; break;
            case 4: wf.y = v; break; case 5: wr.x = v; break; case 6: wr.y = v; break; case 7: wy.y = v; break;
            case 8: xy.x = v; break; case 9: xy.y = v; break; case 10: axy.x = v; break; case 11: axy.y = v; break;
            case 12: cuv.x = v; break; case 13: cuv.y = v; break; case 14: cwy.x = v; break; case 15: cwf.x = v; break;
            case 16: cwf.y = v; break; case 17: cwr.x = v; break; case 18: cwr.y = v; break; case 19: cwy.y = v; break;
            case 20: cxy.x = v; break; default: cxy.y = v; break;
            }
        }
    };
    // COMP: the state update s <- s + h/6 acc as a compensated (Kahan) sum per pair -- y = h/6 acc - c, t = s + y,
    // c = (t - s) - y, s = t: what the addition rounds away is carried in c and given back at the next step (three
    // more packed instructions per pair).  At |x| ~ 100 m the plain sum loses 4e-6 m per step to rounding; this is
    // BASELINE's "fp32 max-abs state error".
    template <bool K2, bool CS, int PRE = 0, int FITSRC = 1, bool COMP = false>   // FITSRC: fp64 only
    __device__ __forceinline__ void advance_state(const DevParams<float> &P, State &X, const float delta[4],
                                                  const float tq[4], const float mu[4], float h, float sd0 = 0.0f,
                                                  float cd0 = 1.0f) const
    {
        State5 S, A;
        S.uv = X.uv; S.wy = X.wy; S.wf = X.wf; S.wr = X.wr; S.xy = X.xy;
        f2 axy_n;
        const bool ok = rk4_step2<K2, CS, false, PRE>(P, Q, K, S, X.axy, delta, tq, mu, h, A, axy_n, nullptr, nullptr,
                                                      nullptr, nullptr, f2{sd0, cd0});
        const f2 h6 = splat(h * (1.0f / 6.0f));
        // s += h/6 acc (:438) unconditionally, then -- behind one wave-uniform, normally-not-taken branch that the
        // normal path falls through -- the SAFE redo overwrites the lanes that left the FAST range.  (Committing only
        // after the test, with the update in an else-branch, put a second TAKEN branch on the normal path: a lone wave
        // pays tens of cycles for each.  The allocator copies the packed state once per step either way.)
        f2 uv, wy, wf, wr, xy;
        if (COMP) {
#define VDYN_KAHAN(f)                                      \
    {                                                      \
        const f2 y_ = fma2(h6, A.f, -X.c##f);              \
        f = S.f + y_;                                      \
        X.c##f = (f - S.f) - y_;                           \
    }
            VDYN_KAHAN(uv) VDYN_KAHAN(wy) VDYN_KAHAN(wf) VDYN_KAHAN(wr) VDYN_KAHAN(xy)
#undef VDYN_KAHAN
        } else {
            uv = fma2(h6, A.uv, S.uv); wy = fma2(h6, A.wy, S.wy); wf = fma2(h6, A.wf, S.wf); wr = fma2(h6, A.wr, S.wr);
            xy = fma2(h6, A.xy, S.xy);
        }
        if (__builtin_expect(__any(!ok) != 0, 0)) {
            if (!ok) {
                float s[10], sn[10], axn, ayn;
#pragma unroll
                for (int i = 0; i < 10; ++i) s[i] = X.get(i);
                float dl[4] = {delta[0], delta[1], delta[2], delta[3]};
                if (PRE == 2) dl[0] = dl[1] = ::atanf(delta[0]);        // delta[0] is tan(delta)
                rk4_step<float, K2, false, true, CS>(P, s, X.axy.x, X.axy.y, dl, tq, mu, h, sn, axn, ayn, nullptr, nullptr);
                uv = f2{sn[0], sn[1]}; wy = f2{sn[2], sn[7]}; wf = f2{sn[3], sn[4]}; wr = f2{sn[5], sn[6]};
                xy = f2{sn[8], sn[9]};
                axy_n = f2{axn, ayn};
                if (COMP) X.cuv = X.cwy = X.cwf = X.cwr = X.cxy = splat(0.0f);   // the redone step starts a fresh sum
            }
        }
        X.uv = uv; X.wy = wy; X.wf = wf; X.wr = wr; X.xy = xy;
        X.axy = axy_n;
    }
    // the same step on the [10] + 2 scalar form (kernels that read single states between steps)
    template <bool K2, bool CS, int PRE = 0>
    __device__ __forceinline__ void advance(const DevParams<float> &P, float s[10], float &ax, float &ay,
                                            const float delta[4], const float tq[4], const float mu[4],
                                            float h, float sd0 = 0.0f, float cd0 = 1.0f) const
    {
        State X;
        X.uv = f2{s[0], s[1]}; X.wy = f2{s[2], s[7]}; X.wf = f2{s[3], s[4]}; X.wr = f2{s[5], s[6]};
        X.xy = f2{s[8], s[9]}; X.axy = f2{ax, ay};
        advance_state<K2, CS, PRE>(P, X, delta, tq, mu, h, sd0, cd0);
        s[0] = X.uv.x; s[1] = X.uv.y; s[2] = X.wy.x; s[7] = X.wy.y;
        s[3] = X.wf.x; s[4] = X.wf.y; s[5] = X.wr.x; s[6] = X.wr.y;
        s[8] = X.xy.x; s[9] = X.xy.y;
        ax = X.axy.x;
        ay = X.axy.y;
    }
    template <bool K2, bool CS, int FITSRC = 1>                   // FITSRC: fp64 only
    __device__ __forceinline__ void advance_diag(const DevParams<float> &P, float s[10], float &ax, float &ay,
                                                 const float delta[4], const float tq[4], const float mu[4],
                                                 float h, float sd[10], Outputs18<float> &o) const
    {
        State5 S, A, D;
        S.uv = f2{s[0], s[1]};
        S.wy = f2{s[2], s[7]};
        S.wf = f2{s[3], s[4]};
        S.wr = f2{s[5], s[6]};
        S.xy = f2{s[8], s[9]};
        Diag2 g;
        f2 axy_n, FzF, FzR;
        const bool ok = rk4_step2<K2, CS, true>(P, Q, K, S, f2{ax, ay}, delta, tq, mu, h, A, axy_n, &D, &g, &FzF, &FzR);
        const f2 h6 = splat(h * (1.0f / 6.0f));
        const f2 uv = fma2(h6, A.uv, S.uv), wy = fma2(h6, A.wy, S.wy), wf = fma2(h6, A.wf, S.wf),
                 wr = fma2(h6, A.wr, S.wr), xy = fma2(h6, A.xy, S.xy);
        float sn[10], axn = axy_n.x, ayn = axy_n.y;
        sn[0] = uv.x; sn[1] = uv.y; sn[2] = wy.x; sn[7] = wy.y;
        sn[3] = wf.x; sn[4] = wf.y; sn[5] = wr.x; sn[6] = wr.y;
        sn[8] = xy.x; sn[9] = xy.y;
        sd[0] = D.uv.x; sd[1] = D.uv.y; sd[2] = D.wy.x; sd[7] = D.wy.y;
        sd[3] = D.wf.x; sd[4] = D.wf.y; sd[5] = D.wr.x; sd[6] = D.wr.y;
        sd[8] = D.xy.x; sd[9] = D.xy.y;
        o.v[0] = g.fx[0].x; o.v[1] = g.fx[0].y; o.v[2] = g.fx[1].x; o.v[3] = g.fx[1].y;        // :420-423
        o.v[4] = g.fy[0].x; o.v[5] = g.fy[0].y; o.v[6] = g.fy[1].x; o.v[7] = g.fy[1].y;
        o.v[8] = FzF.x; o.v[9] = FzF.y; o.v[10] = FzR.x; o.v[11] = FzR.y;
        o.v[12] = g.s[0].x; o.v[13] = g.s[0].y; o.v[14] = g.s[1].x; o.v[15] = g.s[1].y;
        o.v[16] = g.fxt.x; o.v[17] = g.fyt.x;
        if (__builtin_expect(__any(!ok) != 0, 0)) {
            if (!ok) rk4_step<float, K2, true, true, CS>(P, s, ax, ay, delta, tq, mu, h, sn, axn, ayn, sd, &o);
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) s[i] = sn[i];
        ax = axn;
        ay = ayn;
    }
};

}  // namespace vdyn
