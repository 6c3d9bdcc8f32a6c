// vdyn_quad.hpp -- wheel-parallel form of the RK4 step: FOUR lanes per rollout.
//
// At N <= 16384 rollouts the lane-per-rollout kernel leaves most SIMDs empty and its run
// time is the latency of one wave's serial chain.  80 % of a derivative evaluation is
// per-wheel work (vehicle_model.py:261-373), so a quad of adjacent lanes -- FL, FR, RL, RR --
// each takes one wheel, the three chassis sums (sum Fx, sum Fy, yaw moment; :376-378) are
// formed with two DPP quad-permute adds each, and every lane integrates the chassis states
// redundantly plus its own wheel speed.  A wave's chain is ~2x shorter; the chip does ~2x the
// total work, which is free while it is mostly idle.
//
// All four lanes of a quad compute bitwise-identical chassis values: the quad sum is
// (a+b)+(c+d) in every lane (IEEE addition commutes), and everything else is a function of
// replicated inputs.  The summation ORDER differs from the lane-per-rollout kernel
// (((a+b)+c)+d there), so the two kernels agree to rounding, not bit for bit; the
// wheel-parallel form is therefore opt-in (vdyn_set_option), never chosen behind the
// caller's back.
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_device.hpp"

namespace vdyn {

// quad_perm selectors: lane i reads lane sel[i] of its own quad
constexpr int kQuadSwap1 = 0xB1;  // [1,0,3,2]
constexpr int kQuadSwap2 = 0x4E;  // [2,3,0,1]

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) { return __int_as_float(dpp_i32<CTRL>(__float_as_int(v))); }
template <int CTRL>
__device__ __forceinline__ double dpp(double v)
{
    return __hiloint2double(dpp_i32<CTRL>(__double2hiint(v)), dpp_i32<CTRL>(__double2loint(v)));
}

template <typename T>
__device__ __forceinline__ T quad_sum(T v)
{
    v = v + dpp<kQuadSwap1>(v);
    return v + dpp<kQuadSwap2>(v);
}
__device__ __forceinline__ bool quad_all(bool b)
{
    int v = b ? 1 : 0;
    v &= dpp_i32<kQuadSwap1>(v);
    v &= dpp_i32<kQuadSwap2>(v);
    return v != 0;
}

// Per-lane wheel constants (wheel q: 0 FL, 1 FR, 2 RL, 3 RR).
template <typename T>
struct WheelLane {
    T side;      // -1 left, +1 right               (vehicle_model.py:261-271, quirk Q8)
    T lever;     // +a front, -b rear               (:262,:268,:378)
    T B, invB, C;
    T Fz0, kx, ky;  // Fz = Fz0 + kx ax_prev + ky ay_prev   (:255-258)
    bool front;
};

// Wheel q's entry of a per-wheel constant.  The four values are made opaque first: left visible, the select chain
// over P.X[0..3] is folded into ONE load at a lane-dependent index, and an indexed read of a by-value kernel argument
// needs the argument in memory -- the whole DevParams struct was copied to scratch in the prologue of every
// wheel-parallel kernel (264 B fp32 / 392 B fp64 per lane; profiles/r02f: config2_f64_wheel wrote 17 x its
// algorithmic bytes).  Opaque scalars stay in SGPRs and the chain stays three v_cndmask.
template <typename T>
__device__ __forceinline__ T pick_wheel(T a, T b, T c, T d, int q)
{
    asm("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
    return q == 0 ? a : q == 1 ? b : q == 2 ? c : d;
}

template <typename T>
__device__ __forceinline__ WheelLane<T> make_wheel_lane(const DevParams<T> &P, int q)
{
    WheelLane<T> w;
    const bool right = (q & 1) != 0;
    w.front = q < 2;
    w.side = right ? T(1) : T(-1);
    w.lever = w.front ? P.a : -P.b;
    w.B = pick_wheel(P.B[0], P.B[1], P.B[2], P.B[3], q);
    w.invB = pick_wheel(P.invB[0], P.invB[1], P.invB[2], P.invB[3], q);
    w.C = pick_wheel(P.C[0], P.C[1], P.C[2], P.C[3], q);
    w.Fz0 = w.front ? P.Fz0F : P.Fz0R;
    const T dfx = right ? P.DfzxR : P.DfzxL;
    w.kx = w.front ? -dfx : dfx;
    w.ky = w.side * (w.front ? P.DfzyF : P.DfzyR);
    return w;
}

// Lane state: the chassis part replicated in the quad + this lane's wheel speed.
template <typename T>
struct QuadState {
    T U, V, wz, w, yaw, x, y;
};

// One derivative evaluation (vehicle_model.py:220-425) spread over a quad.
// fit: this lane's wheel's fit coefficients (fp64 CS; QuadEngine keeps them in registers), highest degree first.
template <typename T, bool SAFE, bool CS>
__device__ __forceinline__ void planar_deriv_quad(const DevParams<T> &P, const WheelLane<T> &L, const T *fit, T cd, T sd,
                                                  T muFz, T tq, const QuadState<T> &s, T sy, T cy,
                                                  QuadState<T> &k, T &axc, T &ayc)
{
    const T hTw = P.half_T * s.wz;
    const T vxc = fma_t(L.side, hTw, s.U);          // U -+ T wz / 2
    const T vyc = fma_t(L.lever, s.wz, s.V);        // V + a wz | V - b wz
    T fx, fy, fxt, fyt, slip;
    tire_force<T, true, SAFE, CS>(L.B, L.invB, L.C, fit, 1, P.rw, vxc, vyc, s.w, cd, sd,
                                  muFz, fx, fy, fxt, fyt, slip);
    const T Sfx = quad_sum(fx);
    const T Sfy = quad_sum(fy);
    const T Mz = quad_sum(fma_t(L.lever, fy, L.side * P.half_T * fx));   // :378
    k.U = P.inv_m * Sfx + s.V * s.wz;               // :376
    k.V = P.inv_m * Sfy - s.U * s.wz;               // :377
    k.wz = P.inv_Izz * Mz;
    const T fw = L.front ? fxt : fx;                // quirk Q2 (:379-382)
    k.w = (tq - P.rw * fw) * P.inv_Jw;
    k.yaw = s.wz;
    k.x = s.U * cy - s.V * sy;
    k.y = s.U * sy + s.V * cy;
    axc = k.U - s.V * s.wz;
    ayc = k.V + s.U * s.wz;
}

// RK4 (vehicle_model.py:427-445) for one lane of a quad.  Returns false when the lane left the
// validated range of the FAST path.
template <typename T, bool SAFE, bool CS>
__device__ __forceinline__ bool rk4_step_quad(const DevParams<T> &P, const WheelLane<T> &L, const T *fw,
                                              const QuadState<T> &s, T ax, T ay, T delta, T tq, T mu, T h,
                                              QuadState<T> &sn, T &axn, T &ayn)
{
    using M = Math<T, SAFE>;
    bool ok = true;
    T cd, sd;
    M::sincos(delta, &sd, &cd, ok);
    const T Fz = fma_t(L.ky, ay, fma_t(L.kx, ax, L.Fz0));    // :255-258 (quirk Q3)
    const T muFz = mu * Fz;                                  // quirk Q1
    const T hh = T(0.5) * h;
    T sy0, cy0, sy, cy, a1, a2, asx, asy;
    M::sincos(s.yaw, &sy0, &cy0, ok);
    QuadState<T> k, acc, st;

#define VDYN_Q_EACH(OP) OP(U) OP(V) OP(wz) OP(w) OP(yaw) OP(x) OP(y)
    planar_deriv_quad<T, SAFE, CS>(P, L, fw, cd, sd, muFz, tq, s, sy0, cy0, k, a1, a2);        // K1
    asx = a1; asy = a2;
#define VDYN_Q_1(f) acc.f = k.f; st.f = fma_t(hh, k.f, s.f);
    VDYN_Q_EACH(VDYN_Q_1)
    M::stage_sincos(sy0, cy0, st.yaw, hh * k.yaw, &sy, &cy, ok);
    planar_deriv_quad<T, SAFE, CS>(P, L, fw, cd, sd, muFz, tq, st, sy, cy, k, a1, a2);         // K2
    asx += T(2) * a1; asy += T(2) * a2;
#define VDYN_Q_2(f) acc.f = fma_t(T(2), k.f, acc.f); st.f = fma_t(hh, k.f, s.f);
    VDYN_Q_EACH(VDYN_Q_2)
    M::stage_sincos(sy0, cy0, st.yaw, hh * k.yaw, &sy, &cy, ok);
    planar_deriv_quad<T, SAFE, CS>(P, L, fw, cd, sd, muFz, tq, st, sy, cy, k, a1, a2);         // K3
    asx += T(2) * a1; asy += T(2) * a2;
#define VDYN_Q_3(f) acc.f = fma_t(T(2), k.f, acc.f); st.f = fma_t(h, k.f, s.f);
    VDYN_Q_EACH(VDYN_Q_3)
    M::stage_sincos(sy0, cy0, st.yaw, h * k.yaw, &sy, &cy, ok);
    planar_deriv_quad<T, SAFE, CS>(P, L, fw, cd, sd, muFz, tq, st, sy, cy, k, a1, a2);         // K4
    asx += a1; asy += a2;
    const T h6 = h * T(1.0 / 6.0), sixth = T(1.0 / 6.0);
#define VDYN_Q_4(f) sn.f = fma_t(h6, acc.f + k.f, s.f);
    VDYN_Q_EACH(VDYN_Q_4)
#undef VDYN_Q_1
#undef VDYN_Q_2
#undef VDYN_Q_3
#undef VDYN_Q_4
#undef VDYN_Q_EACH
    axn = asx * sixth;
    ayn = asy * sixth;
    return ok;
}

// FAST step, then SAFE for whole quads in which any lane left the validated range.
// (The fp64 lane kernels run their redo under the full exec mask with a select per value: rk4_advance, vdyn_device.hpp.
// The same form HERE crashes this compiler -- ROCm 7.2's clang-22 segfaults in the greedy register allocator,
// VirtRegAuxInfo::isRematerializable, on rollout_quad_kernel<double, 2, 0, false, true> -- so the wheel-parallel
// kernels keep the divergent region; tests/test_isa_audit.py holds every shipped instance, these included, to "no
// vector instruction in front of an exec restore at a join".)
template <typename T, bool CS>
__device__ __forceinline__ void rk4_advance_quad(const DevParams<T> &P, const WheelLane<T> &L, const T *fw,
                                                 QuadState<T> &s, T &ax, T &ay, T delta, T tq, T mu, T h)
{
    QuadState<T> sn;
    T axn, ayn;
    const bool ok = rk4_step_quad<T, false, CS>(P, L, fw, s, ax, ay, delta, tq, mu, h, sn, axn, ayn);
    if (Math<T, false>::kHasRangeLimit) {
        const bool okq = quad_all(ok);
        if (__builtin_expect(__any(!okq) != 0, 0)) {
            if (!okq) rk4_step_quad<T, true, CS>(P, L, fw, s, ax, ay, delta, tq, mu, h, sn, axn, ayn);
        }
    }
    s = sn;
    ax = axn;
    ay = ayn;
}

}  // namespace vdyn
