// vdyn_quad_packed.hpp -- the wheel-parallel fp32 FAST step (vdyn_quad.hpp) on packed pairs.
//
// One lane = one wheel, so there is no second wheel to pair up with; what pairs up are the 2-D
// quantities of the lane itself: corner velocity (vx, vy), slip (sx, sy), tire force (fx, fy),
// (U, V), (x, y), (wz, yaw), (sin, cos), (axc, ayc).  Rotations and cross products are one or two
// VOP3P instructions each (pk_cross / pk_hi_conj / pk_rot90 with the per-half sign bits), the
// tire chain is the lane kernel's fitted one (pacejka_g2x2 in vdyn_packed.hpp, here scalar: one wheel, its own
// nine coefficients), and the integrator state is three pairs + the lane's wheel speed.  457 -> 348 (round 1)
// -> ~285 -> ~240 VALU instructions per RK4 step with the trims of the lane kernel (vdyn_packed.hpp: slips
// pre-multiplied by B, fitted shape function, force sums instead of accelerations, stage yaw increments as small
// rotations in the initial-yaw frame, mod-pi sincos of yaw), all of it on a wave's critical path (these kernels run
// where the chip is mostly empty).
//
// Semantics: those of rk4_step_quad<float, false, true> (vdyn_quad.hpp).  Only the CS = true FAST
// step is specialised; CS = false and the SAFE redo use the scalar code unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_packed.hpp"
#include "vdyn_quad.hpp"

namespace vdyn {

__device__ __forceinline__ f2 pk_rot90(f2 a, f2 b)        // (-a.y b.x, a.x b.x): a rotated by +90 deg, times b.x
{
    f2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

template <typename T>
struct QuadEngine {           // fp64: the scalar step; this lane's wheel's seventeen fit coefficients in registers
    T fw[kTireFitDeg64 + 1];
    __device__ __forceinline__ void init(const DevParams<T> &, const WheelLane<T> &, int q)
    {
        // column q of the kernel-argument table (fit_table_kernarg): a per-lane index, so ordinary (vector) loads
        vdyn_fit_table W = fit_table_kernarg();
#pragma unroll
        for (int i = 0; i <= kTireFitDeg64; ++i) fw[i] = (T)W[4 * i + q];
    }
    template <bool CS>
    __device__ __forceinline__ void advance(const DevParams<T> &P, const WheelLane<T> &L, QuadState<T> &s, T &ax,
                                            T &ay, T delta, T tq, T mu, T h) const
    {
        rk4_advance_quad<T, CS>(P, L, fw, s, ax, ay, delta, tq, mu, h);
    }
};

template <>
struct QuadEngine<float> {
    f2 sck[3];        // (sin, cos) kernel coefficients, as in PkConsts
    f2 scp[4];        // (sin, cos) on |r| <= pi/2 in one chain (PkConsts::scp)
    f2 rot_a, rot_b, rot_c;
    f2 lv;            // (side T/2, lever): corner velocity = (U, V) + lv wz   (:261-271)
    f2 inv_m2;
    float neg_rw_Jw, inv_Jw;             // -rw / Jw; 1 / Jw
    float mom_x, mom_y;                  // side T/2 / Izz, lever / Izz: yaw-moment arms of (fx, fy) (:378)
    float fw[kTireFitDeg + 1];           // this wheel's W_C(c), highest degree first (TireFit in vdyn_device.hpp)

    __device__ __forceinline__ void init(const DevParams<float> &P, const WheelLane<float> &L, int q)
    {
        const float sks[3] = {-1.951163867e-04f, 8.332134224e-03f, -1.666665375e-01f};
        const float cks[3] = {2.443367339e-05f, -1.388732577e-03f, 4.166664556e-02f};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            sck[i] = f2{sks[i], cks[i]};
            asm volatile("" : "+v"(sck[i]));
        }
        const float sn[4] = {2.607052693e-06f, -1.981028618e-04f, 8.333077654e-03f, -1.666665971e-01f};
        const float cc[4] = {2.312937249e-05f, -1.385257230e-03f, 4.166342318e-02f, -4.999989867e-01f};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            scp[i] = f2{sn[i], cc[i]};
            asm volatile("" : "+v"(scp[i]));
        }
        rot_a = f2{-1.0f / 6.0f, 1.0f / 24.0f};
        rot_b = f2{1.0f, -0.5f};
        rot_c = f2{0.0f, 1.0f};
        asm volatile("" : "+v"(rot_a), "+v"(rot_b), "+v"(rot_c));
        float hT = P.half_T, im = P.inv_m, iJ = P.inv_Jw, r_w = P.rw, iIz = P.inv_Izz;
        asm("" : "+v"(hT), "+v"(im), "+v"(iJ), "+v"(r_w), "+v"(iIz));     // see PkParams::init on why
        lv = f2{L.side * hT, L.lever};
        mom_x = L.side * hT * iIz;
        mom_y = L.lever * iIz;
        inv_m2 = f2{im, im};
        inv_Jw = iJ;
        neg_rw_Jw = -r_w * iJ;
#pragma unroll
        for (int i = 0; i <= kTireFitDeg; ++i) fw[i] = pick_wheel(P.W[i][0], P.W[i][1], P.W[i][2], P.W[i][3], q);
    }

    __device__ __forceinline__ f2 sincos_k(float r) const
    {
        const float u = r * r;
        const f2 u2 = f2{u, u};
        f2 p = fma2(sck[0], u2, sck[1]);
        p = fma2(p, u2, sck[2]);
        return fma2(f2{r, u} * u2, p, f2{r, ::fmaf(-0.5f, u, 1.0f)});
    }
    // (sin x, cos x) of the unwrapped yaw: x = k pi + r, one packed chain (sincos_mid2 of vdyn_packed.hpp)
    __device__ __forceinline__ f2 sincos_full(float x, bool &ok) const
    {
        const float k = __builtin_rintf(x * 0.318309886183790671538f);
        float r = ::fmaf(-k, 3.1415927410125732421875f, x);
        r = ::fmaf(-k, -8.74227800037248566e-08f, r);
        const float u = r * r;
        const f2 u2 = f2{u, u};
        f2 p = fma2(scp[0], u2, scp[1]);
        p = fma2(p, u2, scp[2]);
        p = fma2(p, u2, scp[3]);
        const f2 sc = fma2(f2{r * u, u}, p, f2{r, 9.999999404e-01f});
        const unsigned flip = ((unsigned)(int)k) << 31;
        ok = ok && (::fabsf(x) <= fm::kSincosMidLimit);
        return f2{__uint_as_float(__float_as_uint(sc.x) ^ flip), __uint_as_float(__float_as_uint(sc.y) ^ flip)};
    }
    __device__ __forceinline__ f2 stage_rot(float d) const                 // stage_rot2 of vdyn_packed.hpp
    {
        const float u = d * d;
        const f2 a = fma2(f2{u, u}, rot_a, rot_b);
        return fma2(f2{d, u}, a, rot_c);
    }

    struct Lane3 {
        f2 uv, wy, xy;    // (U, V), (wz, yaw), (x, y)
        float w;          // this lane's wheel speed
    };

    // One derivative evaluation (vehicle_model.py:220-425) for this lane's wheel + the replicated chassis.
    // sc = (sin, cos) of the stage's yaw INCREMENT (FIRST: zero): k.xy lives in the frame of the initial yaw.
    // sums = (sum Fx, sum Fy): :413-414's axc, ayc are these over m, scaled once by the caller.
    template <bool FIRST>
    __device__ __forceinline__ void deriv(const DevParams<float> &P, const WheelLane<float> &L, f2 dsc, float muFz,
                                          float tqJ, const Lane3 &s, f2 sc, Lane3 &k, f2 &sums) const
    {
        const f2 wz2 = f2{s.wy.x, s.wy.x};
        const f2 vv = fma2(lv, wz2, s.uv);                                // corner velocity, chassis frame
        const f2 tv = fma2(vv, f2{dsc.y, dsc.y}, pk_cross(vv, dsc));      // :274-281 (vx, vy), tire frame
        const float rvxB = fm::rcp(tv.x) * L.B;                           // B / vx
        const float sx = ::fmaf(P.rw, s.w, -tv.x) * rvxB;                 // B s_x   (:284-287)
        const float sy = -tv.y * ::fabsf(rvxB);                           // B s_y   (:290-293, quirk Q4)
        // G(B s) = sin(C atan x) / x = c W_C(c), c = rsq(1 + x^2) (:296-348; quirk Q5 needs no case, see pacejka_g2x2)
        const float cc = fm::rsq(::fmaf(sx, sx, ::fmaf(sy, sy, 1.0f)));
        float p = ::fmaf(fw[0], cc, fw[1]);
#pragma unroll
        for (int i = 2; i <= kTireFitDeg; ++i) p = ::fmaf(p, cc, fw[i]);
        const float g = p * (cc * muFz);                                  // mu / s times Fz over B (B cancels with the scaled slips)
        const f2 ft = f2{sx, sy} * f2{g, g};                              // :351-360 (fxt, fyt)
        const f2 fc = fma2(ft, f2{dsc.y, dsc.y}, pk_rot90(ft, dsc));      // :363-373 (fx, fy), chassis frame
        const float Sfx = quad_sum(fc.x), Sfy = quad_sum(fc.y);
        const float wzdot = quad_sum(::fmaf(mom_y, fc.y, mom_x * fc.x));  // :378, arms pre-divided by Izz
        sums = f2{Sfx, Sfy};
        const f2 cross = pk_cross(s.uv, s.wy);                            // (V wz, -U wz)
        k.uv = fma2(inv_m2, sums, cross);                                 // :376-377
        k.wy = f2{wzdot, s.wy.x};
        k.w = ::fmaf(neg_rw_Jw, L.front ? ft.x : fc.x, tqJ);              // :379-382, quirk Q2
        if (FIRST) k.xy = s.uv;
        else k.xy = fma2(f2{s.uv.x, s.uv.x}, f2{sc.y, sc.x}, pk_hi_conj(s.uv, sc));   // :384-385, initial-yaw frame
    }

    // FAST RK4 step (vehicle_model.py:427-445); false when the lane left the validated range.
    __device__ __forceinline__ bool step(const DevParams<float> &P, const WheelLane<float> &L, const Lane3 &s, f2 axy,
                                         float delta, float tq, float mu, float h, Lane3 &sn, f2 &axy_n) const
    {
        bool ok = ::fabsf(delta) <= fm::kSincosKernelLimit;
        const f2 dsc = sincos_k(delta);                                   // (sin, cos) of the steering angle
        const float Fz = ::fmaf(L.ky, axy.y, ::fmaf(L.kx, axy.x, L.Fz0)); // :255-258 (quirk Q3)
        const float muFz = mu * Fz, tqJ = tq * inv_Jw;
        const float hh = 0.5f * h;
        const f2 hh2 = f2{hh, hh}, h2 = f2{h, h}, two = f2{2.0f, 2.0f};
        const f2 sc0 = sincos_full(s.wy.y, ok);
        Lane3 k, acc, st;
        f2 a, as2;
        float d2, d3, d4;
#define VDYN_L2_EACH(OP) OP(uv) OP(wy)
        deriv<true>(P, L, dsc, muFz, tqJ, s, sc0, k, a);                  // K1
        as2 = a;
#define VDYN_L3_1(f) acc.f = k.f; st.f = fma2(hh2, k.f, s.f);
        VDYN_L2_EACH(VDYN_L3_1)
        acc.xy = k.xy;
        acc.w = k.w; st.w = ::fmaf(hh, k.w, s.w);
        d2 = hh * k.wy.y;
        deriv<false>(P, L, dsc, muFz, tqJ, st, stage_rot(d2), k, a);      // K2
        as2 = fma2(two, a, as2);
#define VDYN_L3_2(f) acc.f = fma2(two, k.f, acc.f); st.f = fma2(hh2, k.f, s.f);
        VDYN_L2_EACH(VDYN_L3_2)
        acc.xy = fma2(two, k.xy, acc.xy);
        acc.w = ::fmaf(2.0f, k.w, acc.w); st.w = ::fmaf(hh, k.w, s.w);
        d3 = hh * k.wy.y;
        deriv<false>(P, L, dsc, muFz, tqJ, st, stage_rot(d3), k, a);      // K3
        as2 = fma2(two, a, as2);
#define VDYN_L3_3(f) acc.f = fma2(two, k.f, acc.f); st.f = fma2(h2, k.f, s.f);
        VDYN_L2_EACH(VDYN_L3_3)
        acc.xy = fma2(two, k.xy, acc.xy);
        acc.w = ::fmaf(2.0f, k.w, acc.w); st.w = ::fmaf(h, k.w, s.w);
        d4 = h * k.wy.y;
        deriv<false>(P, L, dsc, muFz, tqJ, st, stage_rot(d4), k, a);      // K4
        as2 = as2 + a;
        ok = ok && (::fmaxf(::fmaxf(::fabsf(d2), ::fabsf(d3)), ::fabsf(d4)) <= kStageYawLimit);
        const float sixth = 1.0f / 6.0f, h6 = h * sixth;
        const f2 h62 = f2{h6, h6};
#define VDYN_L3_4(f) sn.f = fma2(h62, acc.f + k.f, s.f);
        VDYN_L2_EACH(VDYN_L3_4)
        {
            const f2 b = acc.xy + k.xy;                                    // initial-yaw frame -> global frame
            const f2 g = fma2(f2{b.x, b.x}, f2{sc0.y, sc0.x}, pk_hi_conj(b, sc0));
            sn.xy = fma2(h62, g, s.xy);
        }
        sn.w = ::fmaf(h6, acc.w + k.w, s.w);
#undef VDYN_L3_1
#undef VDYN_L3_2
#undef VDYN_L3_3
#undef VDYN_L3_4
#undef VDYN_L2_EACH
        axy_n = as2 * (inv_m2 * f2{sixth, sixth});
        return ok;
    }

    template <bool CS>
    __device__ __forceinline__ void advance(const DevParams<float> &P, const WheelLane<float> &L,
                                            QuadState<float> &s, float &ax, float &ay, float delta, float tq,
                                            float mu, float h) const
    {
        if (!CS) {
            rk4_advance_quad<float, CS>(P, L, nullptr, s, ax, ay, delta, tq, mu, h);
            return;
        }
        Lane3 S, Sn;
        S.uv = f2{s.U, s.V};
        S.wy = f2{s.wz, s.yaw};
        S.xy = f2{s.x, s.y};
        S.w = s.w;
        f2 axy_n;
        const bool ok = step(P, L, S, f2{ax, ay}, delta, tq, mu, h, Sn, axy_n);
        QuadState<float> sn;
        sn.U = Sn.uv.x; sn.V = Sn.uv.y; sn.wz = Sn.wy.x; sn.yaw = Sn.wy.y; sn.x = Sn.xy.x; sn.y = Sn.xy.y;
        sn.w = Sn.w;
        float axn = axy_n.x, ayn = axy_n.y;
        const bool okq = quad_all(ok);                                    // the whole quad redoes the step together
        if (__builtin_expect(__any(!okq) != 0, 0)) {
            if (!okq) rk4_step_quad<float, true, CS>(P, L, nullptr, s, ax, ay, delta, tq, mu, h, sn, axn, ayn);
        }
        s = sn;
        ax = axn;
        ay = ayn;
    }
};

}  // namespace vdyn
