// vdyn_controls.hpp -- the controllers either side of the RK4 path, per lane, for gfx950.
//
// Semantics:
//   /root/reference/libs/controllers/stanley_controller.py:56-76    get_lookahead_index
//   /root/reference/libs/controllers/stanley_controller.py:78-129   stanley_control
//   /root/reference/libs/controllers/stanley_controller.py:138-159  long_control
//   /root/reference/libs/vehicle_model/drive.py:128-138             10:1 hold + steering filter
//
// The Stanley controller's nearest-waypoint search is a global scan with "first
// minimum wins" (strict '<' on sqrt distances).  It is kept global and exact here: the
// scan compares squared distances (no sqrt in the loop) and falls back to comparing
// the rounded sqrt values only for a candidate so close to the running minimum that
// the two roundings could coincide.  The controllers run once per `ctrl_every` steps, so
// their trigonometry uses the full-range ROCm device library.
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_device.hpp"

namespace vdyn {

template <typename T>
struct CtrlGains {
    T k, k_soft, max_steer, lookahead, deadband;  // stanley_controller.py:40-47
    T kp, ki, kd;                                 // :140-143
    T filt_keep, filt_gain;                       // drive.py:137: (1 - 1e-5/(2*0.001)), 1e-5/(2*0.001)
};

template <typename T> struct Lib;
// fp32: the library's atan2f / atanf / sincosf / fmodf cost 60-150 instructions each (full-range
// reductions, special-case branches); the controllers call six of them per update.  The fp32 path is
// held to 1e-3 on states (north_star), so it takes the bounded-cost forms of vdyn_fastmath.hpp (2 ulp)
// where the argument allows and the library only beyond (|yaw| > 2^16).  fp64 stays on the library.
template <> struct Lib<float> {
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float atan(float x) { return fm::atan_rcp(x, fm::rcp(x)); }
    static __device__ __forceinline__ float atan2(float y, float x)
    {
        const float ax = ::fabsf(x), ay = ::fabsf(y);
        const float mx = ::fmaxf(ax, ay), mn = ::fminf(ax, ay);
        float t = mn * fm::rcp(mx);
        t = mx > 0.0f ? t : 0.0f;                                   // atan2(0, 0) = 0
        t = (ax == ay && ax > 3.0e38f) ? 1.0f : t;                  // inf / inf
        float p = fm::atan_rcp(t, 1.0f);                            // t in [0, 1]: the |x| > 1 branch is never taken
        p = ay > ax ? 1.57079637050628662109375f - p : p;
        p = x < 0.0f || (x == 0.0f && ::signbit(x)) ? 3.1415927410125732421875f - p : p;
        p = (x != x || y != y) ? ::nanf("") : p;
        return ::copysignf(p, y);
    }
    static __device__ __forceinline__ void sincos(float x, float *s, float *c)
    {
        if (::fabsf(x) <= fm::kSincosMidLimit) fm::sincos_mid(x, s, c);
        else ::sincosf(x, s, c);
    }
    static __device__ __forceinline__ float div(float a, float b) { return a * fm::rcp(b); }
    // (e + pi) mod 2 pi - pi, floor-mod, for |e| up to ~1e5: two-term Cody-Waite instead of fmodf
    static __device__ __forceinline__ float wrap_pi(float e)
    {
        const float pi = 3.1415927410125732421875f;
        if (!(::fabsf(e) <= 2.0e5f)) return wrap_pi_lib(e);
        const float m = e + pi;
        const float k = ::floorf(m * 0.159154943091895335769f);
        float r = ::fmaf(-k, 6.283185482025146484375f, m);
        r = ::fmaf(-k, -1.74845560007449713e-07f, r);
        r = r < 0.0f ? r + 6.283185482025146484375f : r;            // rounding at the seam
        r = r >= 6.283185482025146484375f ? r - 6.283185482025146484375f : r;
        return r - pi;
    }
    static __device__ __forceinline__ float wrap_pi_lib(float e)
    {
        const float pi = 3.141592653589793f, two_pi = 6.283185307179586f;
        float m = ::fmodf(e + pi, two_pi);
        if (m < 0.0f) m += two_pi;
        return m - pi;
    }
    static constexpr float kTieBand = 1.0f - 16.0f * 1.1920929e-07f;
};
template <> struct Lib<double> {
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double atan2(double y, double x) { return ::atan2(y, x); }
    static __device__ __forceinline__ double atan(double x) { return ::atan(x); }
    static __device__ __forceinline__ void sincos(double x, double *s, double *c) { ::sincos(x, s, c); }
    static __device__ __forceinline__ double div(double a, double b) { return a / b; }
    // (e + pi) % (2 pi) - pi with Python's floor-mod (stanley_controller.py:103,123)
    static __device__ __forceinline__ double wrap_pi(double e)
    {
        const double pi = 3.141592653589793, two_pi = 2 * 3.141592653589793;
        double m = ::fmod(e + pi, two_pi);
        if (m < 0.0) m += two_pi;
        return m - pi;
    }
    static constexpr double kTieBand = 1.0 - 16.0 * 2.220446049250313e-16;
};

template <typename T>
__device__ __forceinline__ T wrap_pi(T e) { return Lib<T>::wrap_pi(e); }

// Waypoint access: a table of (x, y) pairs, either staged in LDS or read through L2, and
// (optionally) the table of segment lengths seg[i] = |wp[i] - wp[i-1]| (seg[0] unused) that
// the workgroup computed once, so that the lookahead walk does not take a sqrt per lane
// per waypoint.
constexpr int kWpBlock = 32;  // waypoints per bounding circle

template <typename T>
struct Waypoints {
    const T *base;    // this lane's table: [W][2]
    const T *seg;     // nullable: segment lengths of the same table, seg[i * ss]
    const T *bounds;  // nullable: bounding circles, component c of block b at bounds[(4 * b + c) * bs]:
                      //           centre x, y, radius (inflated), unused
    int W;
    int ss, bs;       // element strides: 1 for a table staged in LDS; the number of tables when the
                      // auxiliary arrays live in global memory TRANSPOSED ([i][P], [b][4][P]), so that
                      // the lanes of a wave -- vehicles with consecutive tables -- read neighbouring words
    __device__ __forceinline__ void get(int i, T &x, T &y) const
    {
        x = base[2 * i];
        y = base[2 * i + 1];
    }
    __device__ __forceinline__ T seg_at(int i) const { return seg[(int64_t)i * ss]; }
    __device__ __forceinline__ void bound(int b, T &cx, T &cy, T &r) const
    {
        cx = bounds[(int64_t)(4 * b) * bs];
        cy = bounds[(int64_t)(4 * b + 1) * bs];
        r = bounds[(int64_t)(4 * b + 2) * bs];
    }
};

template <typename T>
__device__ __forceinline__ T segment_length(T ax, T ay, T bx, T by)
{
    const T ex = bx - ax, ey = by - ay;
    return Lib<T>::sqrt(ex * ex + ey * ey);
}

// stanley_controller.py:56-66: global nearest waypoint, first minimum under strict '<' of
// the sqrt distances.  EXACT = false compares squared distances only (branch-free, so the
// loop unrolls and its LDS reads pipeline) and reports through `ambiguous` whether some
// candidate came so close to the running minimum that the rounded roots might have tied;
// EXACT = true settles every such candidate with the roots themselves.
template <typename T, bool EXACT>
__device__ __forceinline__ void nearest_waypoint(const Waypoints<T> &wp, T x, T y, T &best_d2, int &best_i,
                                                 bool &ambiguous)
{
    using L = Lib<T>;
    best_d2 = T(INFINITY);
    best_i = 0;
    ambiguous = false;
#pragma unroll 8
    for (int i = 0; i < wp.W; ++i) {
        T wx, wy;
        wp.get(i, wx, wy);
        const T dx = wx - x, dy = wy - y;
        const T d2 = dx * dx + dy * dy;
        bool better = d2 < best_d2;
        // sqrt is monotone, so d2 < best_d2 implies sqrt(d2) <= sqrt(best_d2); equality of the
        // rounded roots (the reference would then keep the earlier index) needs d2 within a
        // few ulp of best_d2
        const bool close = better && d2 >= best_d2 * L::kTieBand;
        if (EXACT) {
            if (close) better = L::sqrt(d2) < L::sqrt(best_d2);
        } else {
            ambiguous = ambiguous || close;
        }
        best_d2 = better ? d2 : best_d2;
        best_i = better ? i : best_i;
    }
}

// One block of the scan above (waypoints [lo, hi)), continuing a running minimum.
template <typename T, bool EXACT>
__device__ __forceinline__ void nearest_in_range(const Waypoints<T> &wp, int lo, int hi, T x, T y, T &best_d2,
                                                 int &best_i, bool &ambiguous)
{
    using L = Lib<T>;
#pragma unroll 8
    for (int i = lo; i < hi; ++i) {
        T wx, wy;
        wp.get(i, wx, wy);
        const T dx = wx - x, dy = wy - y;
        const T d2 = dx * dx + dy * dy;
        bool better = d2 < best_d2;
        const bool close = better && d2 >= best_d2 * L::kTieBand;
        if (EXACT) {
            if (close) better = L::sqrt(d2) < L::sqrt(best_d2);
        } else {
            ambiguous = ambiguous || close;
        }
        best_d2 = better ? d2 : best_d2;
        best_i = better ? i : best_i;
    }
}

// The same global search, exactly, with most of the table skipped.  Every 32 consecutive waypoints
// carry a bounding circle (centre c, radius r >= every member's distance to c), so block b holds a
// point within |q - c_b| + r_b of the query q and none nearer than |q - c_b| - r_b.
//   pass 1: U = min_b (|q - c_b| + r_b), an upper bound of the minimum distance;
//   pass 2: a block with |q - c_b| - r_b > U can hold neither the minimum nor a tie with it; the lane
//           keeps [lo, hi], the first and last block that might;
//   scan:   the reference's sequential scan (strict '<', first minimum wins) over the waypoints of
//           blocks lo..hi only -- everything outside is strictly farther than the minimum, so the
//           result is the one of the full scan.
// Each lane scans ITS OWN range (typically 2-3 blocks around its vehicle): lanes of a wave that sit
// at different places along the path do not pay for each other's blocks.  The bounds are read
// eight at a time (one LDS / memory latency per eight circles); every inequality is slackened by
// 1e-6 so that rounding can only widen the range (when in doubt, scan).
// `hint` >= 0: any waypoint index (the nearest one of the previous controller update is the useful
// choice: the vehicle has moved centimetres since).  |q - wp[hint]| is an upper bound of the minimum
// distance too, and a far tighter one than pass 1 finds, so pass 1 is skipped.  The hint only prunes:
// the result is the global first minimum either way.
template <typename T, bool EXACT>
__device__ __forceinline__ void nearest_waypoint_pruned(const Waypoints<T> &wp, T x, T y, T &best_d2, int &best_i,
                                                        bool &ambiguous, int hint = -1)
{
    best_d2 = T(INFINITY);
    best_i = 0;
    ambiguous = false;
    const int nb = (wp.W + kWpBlock - 1) / kWpBlock;
    constexpr int kChunk = 8;
    T U = T(INFINITY);
    const bool hinted = __all(hint >= 0) != 0;                   // wave-uniform: every lane brought a hint
    if (hinted) {
        T hx, hy;
        wp.get(min(hint, wp.W - 1), hx, hy);
        const T ex = hx - x, ey = hy - y;
        U = (T)__builtin_amdgcn_sqrtf((float)(ex * ex + ey * ey)) * T(1.000001) + T(1e-18);
        U = U == U ? U : T(INFINITY);                            // NaN coordinates: no bound
    }
    for (int b0 = 0; b0 < (hinted ? 0 : nb); b0 += kChunk) {
        T ub[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            T cx, cy, r;
            wp.bound(min(b0 + j, nb - 1), cx, cy, r);
            const T ex = cx - x, ey = cy - y;
            // v_sqrt_f32 (1 ulp, no denormal fix-up), inflated: an upper bound does not need the last bits
            // (+1e-18: distances below the float range's root would otherwise round to zero)
            ub[j] = (T)__builtin_amdgcn_sqrtf((float)(ex * ex + ey * ey)) * T(1.000001) + r + T(1e-18);
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j) U = ub[j] < U ? ub[j] : U;     // NaN never lowers U
    }
    int lo = nb, hi = -1;
    for (int b0 = 0; b0 < nb; b0 += kChunk) {
        T dq2[kChunk], rr[kChunk];
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            T cx, cy;
            wp.bound(min(b0 + j, nb - 1), cx, cy, rr[j]);
            const T ex = cx - x, ey = cy - y;
            dq2[j] = (ex * ex + ey * ey) * T(0.999998);
        }
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
            const int b = b0 + j;
            const T reach = U + rr[j];
            const bool need = b < nb && !(dq2[j] > reach * reach);     // |q - c| - r <= U, root-free
            lo = need && b < lo ? b : lo;
            hi = need && b > hi ? b : hi;
        }
    }
    if (hi < lo) { lo = 0; hi = nb - 1; }                                // nothing comparable: plain full scan
    nearest_in_range<T, EXACT>(wp, lo * kWpBlock, min((hi + 1) * kWpBlock, wp.W), x, y, best_d2, best_i, ambiguous);
}

// stanley_controller.py:78-129 -> steering angle (limited), target index, crosstrack error
template <typename T>
__device__ __forceinline__ void stanley_control(const CtrlGains<T> &G, const Waypoints<T> &wp, T x, T y, T yaw,
                                                T v, T &steer_out, int &idx_out, T &cte_out, int *near_io = nullptr)
{
    using L = Lib<T>;
    T best_d2;
    int best_i;
    bool amb;
    if (wp.bounds != nullptr) {
        const int hint = near_io != nullptr ? *near_io : -1;
        nearest_waypoint_pruned<T, false>(wp, x, y, best_d2, best_i, amb, hint);
        if (__builtin_expect(__any(amb) != 0, 0)) {      // wave-uniform, practically never taken
            if (amb) nearest_waypoint_pruned<T, true>(wp, x, y, best_d2, best_i, amb, hint);
        }
        if (near_io != nullptr) *near_io = best_i;
    } else {
        nearest_waypoint<T, false>(wp, x, y, best_d2, best_i, amb);
        if (__builtin_expect(__any(amb) != 0, 0)) {
            if (amb) nearest_waypoint<T, true>(wp, x, y, best_d2, best_i, amb);
        }
    }
    // :68-76 walk forward until the accumulated arc length reaches the lookahead distance
    T total = L::sqrt(best_d2);
    int ce = best_i;
    T px, py;
    wp.get(best_i, px, py);
    if (sizeof(T) == 4 && wp.seg != nullptr) {
        // fp32: wp.seg holds the CUMULATIVE arc length cum[i] = sum_{j <= i} |wp[j] - wp[j-1]| (waypoint_cumsum_kernel),
        // and the walk becomes a search for the first i > best_i with total + cum[i] - cum[best_i] >= lookahead:
        // a guess from the table's mean spacing, an 8-entry window around it, a bisection when the window
        // misses.  The reference adds the segments one by one; the two sums differ by fp32 rounding only
        // (167 dependent adds at 5 m / 3 cm: the chain this removes), which can move the index by one exactly
        // where the fp32 sequential sum itself is within rounding of the boundary.  fp64 keeps the exact walk.
        const int last = wp.W - 1;
        if (total < G.lookahead && best_i < last) {
            const T target = wp.seg_at(best_i) + (G.lookahead - total);
            const T span = wp.seg_at(last);
            int lo = best_i, hi = last;                          // invariant: cum[lo] < target; answer in (lo, hi]
            const T per = (T)last / span;                         // waypoints per metre (inf / NaN: the bisection decides)
            T gf = (T)best_i + (G.lookahead - total) * per;
            gf = gf < (T)(best_i + 1) ? (T)(best_i + 1) : gf;
            gf = gf > (T)last ? (T)last : gf;                     // NaN stays NaN -> int conversion clamps below
            int g = (int)gf;
            g = min(max(g, best_i + 1), last);
            constexpr int kWin = 8;
            const int w0 = min(max(g - kWin / 2, best_i + 1), max(last - kWin + 1, best_i + 1));
            T cw[kWin];
#pragma unroll
            for (int k = 0; k < kWin; ++k) cw[k] = wp.seg_at(min(w0 + k, last));
            // window entries below the target raise lo, entries at or above it lower hi (cum is non-decreasing)
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const int i = min(w0 + k, last);
                const bool below = cw[k] < target;
                lo = below && i > lo ? i : lo;
                hi = !below && i < hi && cw[k] == cw[k] ? i : hi;
            }
            while (__any(hi - lo > 1)) {                          // normally zero trips: the window bracketed the crossing
                const int mid = (lo + hi) >> 1;
                const T cm = wp.seg_at(mid);
                const bool below = cm < target;
                const bool act = hi - lo > 1;
                lo = act && below ? mid : lo;
                hi = act && !below ? mid : hi;
            }
            // cum[last] < target (the path ends before the lookahead distance): the reference stops at the last waypoint
            ce = hi;
        }
        wp.get(ce, px, py);
    } else if (wp.seg != nullptr) {
        // same sequential sum as the reference, sixteen precomputed segment lengths per trip.  The
        // running total never decreases (lengths are >= 0), so a trip whose LAST partial sum is
        // still short of the lookahead cannot contain the crossing: sixteen dependent adds and one
        // compare, and only the trip that ends the walk is resolved entry by entry.
        bool done = total >= G.lookahead;
        constexpr int kTrip = 16;
        for (int i0 = best_i + 1; i0 < wp.W && !done; i0 += kTrip) {
            T t[kTrip];
#pragma unroll
            for (int k = 0; k < kTrip; ++k) t[k] = wp.seg_at(min(i0 + k, wp.W - 1));
            t[0] = total + t[0];
#pragma unroll
            for (int k = 1; k < kTrip; ++k) t[k] = t[k - 1] + t[k];
            const bool whole = i0 + kTrip <= wp.W && t[kTrip - 1] < G.lookahead;      // NaN: false -> resolved below
            if (__any(!whole)) {
                if (!whole) {
#pragma unroll
                    for (int k = 0; k < kTrip; ++k) {
                        const bool take = !done && (i0 + k) < wp.W;
                        total = take ? t[k] : total;
                        ce = take ? i0 + k : ce;
                        done = done || !take || total >= G.lookahead;
                    }
                }
            }
            if (whole) {
                total = t[kTrip - 1];
                ce = i0 + kTrip - 1;
            }
        }
        wp.get(ce, px, py);
    } else {
        for (int i = best_i + 1; i < wp.W; ++i) {
            if (total >= G.lookahead) break;
            T qx, qy;
            wp.get(i, qx, qy);
            total += segment_length<T>(px, py, qx, qy);
            ce = i;
            px = qx;
            py = qy;
        }
    }
    // :90-98 (px, py) is waypoint ce
    T sy, cy;
    L::sincos(yaw, &sy, &cy);
    const T v0 = px - x - G.lookahead * cy;
    const T v1 = py - y - G.lookahead * sy;
    T cte = L::sqrt(v0 * v0 + v1 * v1);
    if (cte < G.deadband) cte = T(0);
    // :101-104
    const T che = wrap_pi<T>(L::atan2(v1, v0) - yaw);
    const T sign = che > T(0) ? T(1) : (che < T(0) ? T(-1) : che);  // np.sign: 0 and nan pass through
    // :109-120 trajectory heading; wraps from the last waypoint to the first
    T ax_, ay_, bx_, by_;
    if (ce < wp.W - 1) {
        ax_ = px; ay_ = py;
        wp.get(ce + 1, bx_, by_);
    } else {
        wp.get(wp.W - 1, ax_, ay_);
        wp.get(0, bx_, by_);
    }
    const T he = wrap_pi<T>(L::atan2(by_ - ay_, bx_ - ax_) - yaw);         // :122-123
    T steer = he + L::atan(L::div(G.k * sign * cte, v + G.k_soft));        // :124-126
    steer = steer < -G.max_steer ? -G.max_steer : steer;                   // :128 np.clip
    steer = steer > G.max_steer ? G.max_steer : steer;
    steer_out = steer;
    idx_out = ce;
    cte_out = cte;
}

// stanley_controller.py:138-159
template <typename T>
__device__ __forceinline__ void long_control(const CtrlGains<T> &G, T desired, T current, T prev, T &total, T dt,
                                             T &tau_out)
{
    const T vel_error = desired - current;
    total = total + vel_error * dt;
    const T p = G.kp * vel_error;
    const T i = G.ki * total;
    const T d = G.kd * (current - prev) / dt;
    T tau = p + i + d;
    if (current <= T(0.01)) tau = abs_t(tau);
    tau_out = tau;
}

// Controller state carried between controller updates (and between launches):
// rows of cstate [6][N]: x_del, total_vel_error, prev_vel, target_vel, delta, torque.
template <typename T>
struct CtrlState {
    T x_del, total, prev_vel, target, delta, tau;
    int idx;  // last target index (diagnostic)
    T cte;    // last crosstrack error (diagnostic)
    int near = -1;  // nearest waypoint of the last update: prunes the next search (never changes its result)
};

// One controller update, drive.py:128-138, from the current vehicle state s[10];
// `steer` returns the unfiltered (limited) Stanley angle.
template <typename T>
__device__ __forceinline__ void controller_update(const CtrlGains<T> &G, const Waypoints<T> &wp, const T s[10],
                                                  T dt, CtrlState<T> &c, T &steer)
{
    stanley_control<T>(G, wp, s[8], s[9], s[7], s[0], steer, c.idx, c.cte, &c.near);  // drive.py:129-130
    long_control<T>(G, c.target, s[0], c.prev_vel, c.total, dt, c.tau);       // :131-133
    c.prev_vel = s[0];                                                        // :134
    c.x_del = G.filt_keep * c.x_del + G.filt_gain * steer;                    // :137
    c.delta = c.x_del;                                                        // :138
}

}  // namespace vdyn
