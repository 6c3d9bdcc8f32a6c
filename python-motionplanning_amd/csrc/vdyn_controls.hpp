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

// Diagnostic builds only (tools/ubench/cl_harness.hip defines VDYN_STAMPS): per-phase cycle totals of one wave's
// controller updates, s_memtime deltas accumulated by lane 0 of wave 0 of workgroup 0 into a buffer nothing else reads.
#ifdef VDYN_STAMPS
__device__ unsigned long long g_vdyn_phase[16];
struct PhaseClock {
    unsigned long long t0;
    __device__ __forceinline__ static unsigned long long now()
    {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    }
    __device__ __forceinline__ void start() { t0 = now(); }
    __device__ __forceinline__ void lap(int phase)
    {
        const unsigned long long t = now();
        if (blockIdx.x == 0 && threadIdx.x == 0) g_vdyn_phase[phase] += t - t0;
        t0 = now();
    }
};
#define VDYN_PHASE_START(pc) (pc).start()
#define VDYN_PHASE_LAP(pc, i) (pc).lap(i)
__device__ unsigned long long g_vdyn_event[4];           // counted over ALL waves (one add per wave and event)
#define VDYN_EVENT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_vdyn_event[i], 1ull); } while (0)
#else
struct PhaseClock {};
#define VDYN_PHASE_START(pc) (void)(pc)
#define VDYN_PHASE_LAP(pc, i) (void)(pc)
#define VDYN_EVENT(i) (void)0
#endif

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
    // v_sqrt_f32 (1 ulp, no denormal fix-up): the lookahead's first term and the crosstrack error, both compared
    // against metres; the IEEE form above stays where rounded roots decide an index (nearest_in_subblocks_exact)
    static __device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
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
    static __device__ __forceinline__ double sqrt_fast(double x) { return ::sqrt(x); }
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
constexpr int kSubBlock = 8;  // waypoints per second-level circle (LDS image of the closed loop only)
constexpr int kSubPerBlock = kWpBlock / kSubBlock;

// SOA = false: (x, y) pairs, base[2 i], base[2 i + 1] -- the caller's table, read in place.
// SOA = true:  the closed loop's LDS image (ClosedLoopLds): x and y rows apart (y entry = x entry + yo), rows 16-byte
//              aligned and padded to whole sub-blocks with x = +inf (a padded entry is never the nearest), plus one
//              bounding circle per kSubBlock waypoints -- what the packed block scan of nearest_waypoint_pruned needs.
template <typename T, bool SOA = false>
struct Waypoints {
    static constexpr bool kSoa = SOA;
    const T *base;    // this lane's table: [W][2], or its x row
    const T *seg;     // nullable: segment lengths of the same table, seg[i * ss]
    const T *bounds;  // nullable: bounding circles, component c of block b at bounds[(4 * b + c) * bs]:
                      //           centre x, y, radius (inflated), unused
    const T *sub = nullptr;  // SOA: second-level circles, one per kSubBlock waypoints
    int W;
    int ss, bs;       // element strides: 1 for a table staged in LDS; the number of tables when the
                      // auxiliary arrays live in global memory TRANSPOSED ([i][P], [b][4][P]), so that
                      // the lanes of a wave -- vehicles with consecutive tables -- read neighbouring words
    T per = T(0);     // SOA, fp32: waypoints per metre of this lane's table, (W - 1) / cum[W - 1] (the lookahead's first guess)
    int yo = 1;       // SOA: offset of the y row
    // SOA: both levels of circles are rows too -- centre x at bounds[b] / sub[j], centre y `bo` / `so` elements further,
    // the (inflated) radius twice that; rows are 16-byte aligned and padded with centre x = +inf (never reached) to
    // `nbu` / `nsbu` entries, multiples of 16 that are the same for every lane of the launch
    int bo = 0, so = 0, nbu = 0, nsbu = 0;
    __device__ __forceinline__ void get(int i, T &x, T &y) const
    {
        if (SOA) {
            x = base[i];
            y = base[i + yo];
        } else {
            x = base[2 * i];
            y = base[2 * i + 1];
        }
    }
    __device__ __forceinline__ T seg_at(int i) const { return seg[(int64_t)i * ss]; }
    __device__ __forceinline__ void bound(int b, T &cx, T &cy, T &r) const
    {
        if (SOA) {
            cx = bounds[b];
            cy = bounds[b + bo];
            r = bounds[b + 2 * bo];
        } else {
            cx = bounds[(int64_t)(4 * b) * bs];
            cy = bounds[(int64_t)(4 * b + 1) * bs];
            r = bounds[(int64_t)(4 * b + 2) * bs];
        }
    }
};

// Element offsets (units of T) of the closed loop's LDS image for P tables of at most Wmax waypoints.  Shared by
// the kernel, its launcher and closed_loop_aux_bytes (which layout the aux buffer takes follows from `bytes()`).
template <typename T>
struct ClosedLoopLds {
    int ws;                 // x / y row stride: whole sub-blocks + 4 (rows of different paths start 4 banks apart)
    int nb, nsb;            // bounding circles / second-level circles per table
    int nbu, nsbu;          // ... padded to multiples of 16 (what a lane may read: two chunks of 8 per trip)
    int brs, srs;           // row strides of the circle rows
    int segs;               // row stride of the segment lengths / cumulative arcs (rows of different paths 4 banks apart)
    size_t xs, ys, seg, bnd, sub, total;   // bnd / sub: centre-x rows of all tables, then centre-y rows, then radii
    __host__ __device__ ClosedLoopLds(int P, int Wmax)
    {
        auto up4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
        nb = (Wmax + kWpBlock - 1) / kWpBlock;
        nsb = (Wmax + kSubBlock - 1) / kSubBlock;
        nbu = (nb + 15) & ~15;
        nsbu = (nsb + 15) & ~15;
        brs = nbu + 4;
        srs = nsbu + 20;                                 // + a trip of padding: a lane past its range reads entries nsbu .. nsbu + 15
        ws = nsb * kSubBlock + 4;
        xs = 0;
        ys = xs + (size_t)P * ws;
        segs = (int)up4((size_t)Wmax) + 4;
        seg = ys + (size_t)P * ws;
        bnd = seg + (size_t)P * segs;
        sub = bnd + (size_t)3 * P * brs;
        total = sub + (size_t)3 * P * srs;
    }
    __host__ __device__ size_t bytes() const { return total * sizeof(T); }
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
template <typename T, bool EXACT, typename WP>
__device__ __forceinline__ void nearest_waypoint(const WP &wp, T x, T y, T &best_d2, int &best_i,
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
template <typename T, bool EXACT, typename WP>
__device__ __forceinline__ void nearest_in_range(const WP &wp, int lo, int hi, T x, T y, T &best_d2,
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

// ---- the packed block scan of the LDS image (Waypoints<T, true>) -------------------------------------------
// Squared distances of the eight waypoints of sub-block i0 / 8 to (x, y): two 16-byte LDS reads per row; fp32 on
// packed pairs (two waypoints per v_pk_add / v_pk_mul / v_pk_fma).  Explicit fma so that the scan and the later
// resolve of the winning sub-block see the same eight values bit for bit.
typedef float vdyn_f4 __attribute__((ext_vector_type(4)));
typedef float vdyn_f2 __attribute__((ext_vector_type(2)));
typedef double vdyn_d2 __attribute__((ext_vector_type(2)));

template <typename T> struct Block8;
template <> struct Block8<float> {
    vdyn_f4 xa, xb, ya, yb;
    __device__ __forceinline__ void load(const float *xr, int yo, int i0)
    {
        const vdyn_f4 *px = reinterpret_cast<const vdyn_f4 *>(xr + i0), *py = reinterpret_cast<const vdyn_f4 *>(xr + i0 + yo);
        xa = px[0]; xb = px[1]; ya = py[0]; yb = py[1];
    }
    __device__ __forceinline__ void d2(float x, float y, float out[8]) const
    {
        const vdyn_f2 qx = vdyn_f2{x, x}, qy = vdyn_f2{y, y};
        const vdyn_f2 wx[4] = {xa.xy, xa.zw, xb.xy, xb.zw}, wy[4] = {ya.xy, ya.zw, yb.xy, yb.zw};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const vdyn_f2 dx = wx[k] - qx, dy = wy[k] - qy;
            const vdyn_f2 d = __builtin_elementwise_fma(dy, dy, dx * dx);
            out[2 * k] = d.x;
            out[2 * k + 1] = d.y;
        }
    }
};
template <> struct Block8<double> {
    vdyn_d2 wx[4], wy[4];
    __device__ __forceinline__ void load(const double *xr, int yo, int i0)
    {
        const vdyn_d2 *px = reinterpret_cast<const vdyn_d2 *>(xr + i0), *py = reinterpret_cast<const vdyn_d2 *>(xr + i0 + yo);
#pragma unroll
        for (int k = 0; k < 4; ++k) { wx[k] = px[k]; wy[k] = py[k]; }
    }
    __device__ __forceinline__ void d2(double x, double y, double out[8]) const
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double dx0 = wx[k].x - x, dy0 = wy[k].x - y, dx1 = wx[k].y - x, dy1 = wy[k].y - y;
            out[2 * k] = ::fma(dy0, dy0, dx0 * dx0);
            out[2 * k + 1] = ::fma(dy1, dy1, dx1 * dx1);
        }
    }
};
__device__ __forceinline__ float min_t(float a, float b) { return __builtin_fminf(a, b); }     // a NaN operand is ignored:
__device__ __forceinline__ double min_t(double a, double b) { return __builtin_fmin(a, b); }   // "NaN never wins"

// Four consecutive entries of an LDS row at a 16-byte aligned address: one ds_read_b128 (fp64: two), which the LDS
// serves without bank conflicts for lanes at consecutive quads (MI355X LDS table: 16 lanes x 4 banks per cycle).
__device__ __forceinline__ void load4(const float *p, float v[4])
{
    const vdyn_f4 q = *reinterpret_cast<const vdyn_f4 *>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void load4(const double *p, double v[4])
{
    const vdyn_d2 a = reinterpret_cast<const vdyn_d2 *>(p)[0], b = reinterpret_cast<const vdyn_d2 *>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
}
__device__ __forceinline__ void store4(float *p, const float v[4])
{
    *reinterpret_cast<vdyn_f4 *>(p) = vdyn_f4{v[0], v[1], v[2], v[3]};
}
__device__ __forceinline__ void store4(double *p, const double v[4])
{
    reinterpret_cast<vdyn_d2 *>(p)[0] = vdyn_d2{v[0], v[1]};
    reinterpret_cast<vdyn_d2 *>(p)[1] = vdyn_d2{v[2], v[3]};
}
__device__ __forceinline__ float coord_slack(float) { return 4.8e-7f; }     // 4 ulp of a coordinate, relative
__device__ __forceinline__ double coord_slack(double) { return 8.9e-16; }

// q = j / d, r = j % d for 0 <= j < 2^23 and a divisor d whose reciprocal rd = 1.0f / d the caller keeps: a float
// product and two corrections instead of the ~30-instruction integer division (the image builder divides per entry).
__device__ __forceinline__ void divmod_small(int j, int d, float rd, int &q, int &r)
{
    q = (int)((float)j * rd);
    r = j - q * d;
    const bool under = r < 0, over = r >= d;
    q += over ? 1 : (under ? -1 : 0);
    r += over ? -d : (under ? d : 0);
}

// The closed loop's LDS image (ClosedLoopLds), built by the calling workgroup (BLOCK threads) from the caller's tables
// wp [P][Wmax][2] and wcount [P] alone.  (Two table kernels used to make half of it per call -- 15 us of launches on
// the stream -- and the copy into LDS waited for one global load at a time: together 50 of a call's 250 us.)  Every
// loop keeps several independent loads in flight and divides by multiplication; a 1024-waypoint x 7 image takes
// about 20k cycles.
template <typename T, int BLOCK>
__device__ __forceinline__ void build_closed_loop_lds(T *__restrict__ lds, const ClosedLoopLds<T> &LL,
                                                      const T *__restrict__ wp, int Wmax, const int *__restrict__ wcount,
                                                      int Pn)
{
    const int tid = threadIdx.x;
    const int yo = (int)(LL.ys - LL.xs);
    constexpr int kFly = 8;                                                 // global loads in flight per thread
    PhaseClock pcb;
    VDYN_PHASE_START(pcb);
    // 1. (x, y) pairs -> x rows and y rows (raw: entries past a table's own end are masked in step 4)
    {
        const int total = Pn * Wmax;                                        // waypoints of all tables
        const uintptr_t addr = reinterpret_cast<uintptr_t>(wp);
        if (sizeof(T) == 4 && (Wmax & 1) == 0 && (addr & 15) == 0) {        // two whole pairs per 16-byte load
            const vdyn_f4 *src = reinterpret_cast<const vdyn_f4 *>(wp);
            const int nq = total / 2, half = Wmax / 2;
            const float rhalf = 1.0f / (float)half;
            for (int base = 0; base < nq; base += BLOCK * kFly) {
                vdyn_f4 q[kFly];
#pragma unroll
                for (int u = 0; u < kFly; ++u) q[u] = src[min(base + u * BLOCK + tid, nq - 1)];
#pragma unroll
                for (int u = 0; u < kFly; ++u) {
                    const int j = base + u * BLOCK + tid;
                    int p, k;
                    divmod_small(min(j, nq - 1), half, rhalf, p, k);
                    if (j < nq) {
                        float *dx = reinterpret_cast<float *>(lds) + LL.xs + (size_t)p * LL.ws + 2 * k;
                        *reinterpret_cast<vdyn_f2 *>(dx) = vdyn_f2{q[u].x, q[u].z};
                        *reinterpret_cast<vdyn_f2 *>(dx + yo) = vdyn_f2{q[u].y, q[u].w};
                    }
                }
            }
        } else {
            const float rw = 1.0f / (float)Wmax;
            for (int base = 0; base < total; base += BLOCK * kFly) {
                T qx[kFly], qy[kFly];
#pragma unroll
                for (int u = 0; u < kFly; ++u) {
                    const int j = min(base + u * BLOCK + tid, total - 1);
                    qx[u] = wp[2 * (int64_t)j];
                    qy[u] = wp[2 * (int64_t)j + 1];
                }
#pragma unroll
                for (int u = 0; u < kFly; ++u) {
                    const int j = base + u * BLOCK + tid;
                    int p, i;
                    divmod_small(min(j, total - 1), Wmax, rw, p, i);
                    if (j < total) {
                        lds[LL.xs + (size_t)p * LL.ws + i] = qx[u];
                        lds[LL.ys + (size_t)p * LL.ws + i] = qy[u];
                    }
                }
            }
        }
        const int padw = LL.ws - Wmax;                                      // the rows' padding: never the nearest
        const float rpad = 1.0f / (float)padw;
        for (int i = tid; i < Pn * padw; i += BLOCK) {
            int p, k;
            divmod_small(i, padw, rpad, p, k);
            lds[LL.xs + (size_t)p * LL.ws + Wmax + k] = T(INFINITY);
            lds[LL.ys + (size_t)p * LL.ws + Wmax + k] = T(0);
        }
    }
    __syncthreads();
    VDYN_PHASE_LAP(pcb, 12);
    // 2. segment lengths seg[p][j] = |wp[j] - wp[j - 1]|, seg[p][0] = 0, of ALL Wmax entries -- what waypoint_aux_kernel
    //    writes for the tables that stay in global memory (same function, same values).  Four per trip.
    {
        const int nquad = (Wmax + 3) / 4;
        const float rq = 1.0f / (float)nquad;
        for (int i = tid; i < Pn * nquad; i += BLOCK) {
            int p, g;
            divmod_small(i, nquad, rq, p, g);
            const T *rx = lds + LL.xs + (size_t)p * LL.ws + 4 * g;          // rows are padded: reads past Wmax are in bounds
            T x[5], y[5];
            x[0] = g > 0 ? rx[-1] : T(0);
            y[0] = g > 0 ? rx[yo - 1] : T(0);
            load4(rx, x + 1);
            load4(rx + yo, y + 1);
            T *out = lds + LL.seg + (size_t)p * LL.segs + 4 * g;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (4 * g + k < Wmax) out[k] = 4 * g + k == 0 ? T(0) : segment_length<T>(x[k], y[k], x[k + 1], y[k + 1]);
        }
    }
    VDYN_PHASE_LAP(pcb, 13);
    // 3. one circle per 32 and per 8 waypoints: bounding-box centre, largest member distance (inflated); rows
    //    (centre x | centre y | radius); padding and blocks past a table's end get centre x = +inf (never reached)
    {
        const int bo = Pn * LL.brs, so = Pn * LL.srs;
        const float rb = 1.0f / (float)LL.brs, rs = 1.0f / (float)LL.srs;
        for (int i = tid; i < Pn * (LL.brs + LL.srs); i += BLOCK) {
            const bool coarse = i < Pn * LL.brs;
            const int ii = coarse ? i : i - Pn * LL.brs;
            int p, b;
            divmod_small(ii, coarse ? LL.brs : LL.srs, coarse ? rb : rs, p, b);
            const int len = coarse ? kWpBlock : kSubBlock, nreal = coarse ? LL.nb : LL.nsb;
            const int cnt = b < nreal ? min(max(wcount[p], 1), Wmax) - b * len : 0;      // members: the first min(cnt, len)
            const T *rx = lds + LL.xs + (size_t)p * LL.ws + min(b, nreal - 1) * len;
            // Bounding box of the members.  Neighbouring threads own neighbouring circles, whose rows lie `len` entries
            // apart -- the same LDS bank for all of them -- so every thread starts at a different member and wraps
            // around (min / max do not care about the order): conflict-free reads instead of 32-way conflicts.
            const int rot = coarse ? tid : tid >> 2, lm = len - 1;
            T x0 = T(INFINITY), x1 = -T(INFINITY), y0 = T(INFINITY), y1 = -T(INFINITY);
            for (int k0 = 0; k0 < len; k0 += kSubBlock) {                   // eight members per trip, their reads together
                T wx[kSubBlock], wy[kSubBlock];
                int kk[kSubBlock];
#pragma unroll
                for (int k = 0; k < kSubBlock; ++k) {
                    kk[k] = (k0 + k + rot) & lm;
                    wx[k] = rx[kk[k]];
                    wy[k] = rx[kk[k] + yo];
                }
#pragma unroll
                for (int k = 0; k < kSubBlock; ++k) {
                    const bool in = kk[k] < cnt;
                    x0 = in && wx[k] < x0 ? wx[k] : x0; x1 = in && wx[k] > x1 ? wx[k] : x1;
                    y0 = in && wy[k] < y0 ? wy[k] : y0; y1 = in && wy[k] > y1 ? wy[k] : y1;
                }
            }
            // centre of the box; radius = its half diagonal (every member lies in the box) plus the rounding of the
            // centre itself -- no second pass over the members.  (The tables that stay in global memory keep the
            // largest member distance, waypoint_aux_kernel: the circles only prune, the search result is the same.)
            const T cx = T(0.5) * (x0 + x1), cy = T(0.5) * (y0 + y1);
            const T hx = T(0.5) * (x1 - x0), hy = T(0.5) * (y1 - y0);
            const T rad = (Lib<T>::sqrt(hx * hx + hy * hy) + (abs_t(cx) + abs_t(cy)) * coord_slack(T(0))) * T(1.00001) + T(1e-30);
            const bool empty = cnt <= 0;
            T *row = lds + (coarse ? LL.bnd : LL.sub) + ii;
            const int o = coarse ? bo : so;
            row[0] = empty ? T(INFINITY) : cx;
            row[o] = empty ? T(0) : cy;
            row[2 * o] = empty ? T(0) : rad;                                 // never smaller than the true radius
        }
    }
    __syncthreads();
    VDYN_PHASE_LAP(pcb, 14);
    // 4. entries past a table's own end: x = +inf, never the nearest (the block scan reads whole sub-blocks)
    for (int p = 0; p < Pn; ++p)
        for (int i = min(max(wcount[p], 1), Wmax) + tid; i < Wmax; i += BLOCK) {
            lds[LL.xs + (size_t)p * LL.ws + i] = T(INFINITY);
            lds[LL.ys + (size_t)p * LL.ws + i] = T(0);
        }
    // 5. fp32: cumulative arc length instead of segment lengths (stanley_control), summed exactly as
    //    waypoint_cumsum_kernel sums the tables that stay in global memory: one wave per table, every lane its chunk
    //    in order, a wave scan for the chunk offsets
    if (sizeof(T) == 4) {
        const int lane = tid & 63, chunk = (Wmax + 63) / 64;
        for (int p = tid >> 6; p < Pn; p += BLOCK / 64) {
            T *sg = lds + LL.seg + (size_t)p * LL.segs;
            const int j0 = lane * chunk, j1 = min(j0 + chunk, Wmax);
            T sum = T(0);
            if ((chunk & 3) == 0 && j1 - j0 == chunk) {                     // whole aligned chunk: 16-byte reads (4-way bank
                for (int j = j0; j < j1; j += 8) {                          // conflicts between the lanes instead of 16-way)
                    T v[8];
                    load4(sg + j, v);
                    if (j + 4 < j1) load4(sg + j + 4, v + 4);
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum = j + k < j1 ? sum + v[k] : sum;
                }
            } else {
                for (int j = j0; j < j1; j += 8) {                          // eight reads in flight, added in order
                    T v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = sg[min(j + k, Wmax - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum = j + k < j1 ? sum + v[k] : sum;
                }
            }
            T incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const T o = __shfl_up(incl, off);
                if (lane >= off) incl += o;
            }
            T run = incl - sum;
            if ((chunk & 3) == 0 && j1 - j0 == chunk) {
                for (int j = j0; j < j1; j += 4) {
                    T v[4];
                    load4(sg + j, v);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { run += v[k]; v[k] = run; }
                    store4(sg + j, v);
                }
            } else {
                for (int j = j0; j < j1; j += 8) {
                    T v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = sg[min(j + k, Wmax - 1)];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        run = j + k < j1 ? run + v[k] : run;
                        if (j + k < j1) sg[j + k] = run;
                    }
                }
            }
        }
    }
    __syncthreads();
    VDYN_PHASE_LAP(pcb, 15);
}


// Eight bounding circles at once, rows as Waypoints<T, true> keeps them (centre x at cxr[b0 + j], centre y `o` further,
// radius `2 o` further; b0 a multiple of 4): bit j of the result is set when circle b0 + j lies farther than `U` from
// (x, y) -- it can then hold neither the nearest waypoint nor a tie with it.  The test is the sign of
//     s = (U + r)^2 - (cx - x)^2 - (cy - y)^2,
// root-free; U and r arrive inflated (1e-6 / 1e-5 relative), which is ten times the rounding of s where it changes sign,
// so rounding can only clear a bit (when in doubt, scan).  fp32: two circles per packed instruction, six per pair, and
// one v_alignbit_b32 per circle shifts the sign of s into the mask: 4 VALU instructions per circle (15 in the scalar
// form with compares and selects, which was a quarter of a controller update).
template <typename T> struct Circles8;
template <> struct Circles8<float> {
    vdyn_f4 ca, cb, ya, yb, ra, rb;
    __device__ __forceinline__ void load(const float *cxr, int o, int b0)
    {
        const vdyn_f4 *pc = reinterpret_cast<const vdyn_f4 *>(cxr + b0), *pyy = reinterpret_cast<const vdyn_f4 *>(cxr + b0 + o),
                      *pr = reinterpret_cast<const vdyn_f4 *>(cxr + b0 + 2 * o);
        ca = pc[0]; cb = pc[1]; ya = pyy[0]; yb = pyy[1]; ra = pr[0]; rb = pr[1];
    }
    __device__ __forceinline__ unsigned skip(float x, float y, float U) const
    {
        const vdyn_f2 qx = vdyn_f2{x, x}, qy = vdyn_f2{y, y}, u2 = vdyn_f2{U, U};
        const vdyn_f2 cx[4] = {ca.xy, ca.zw, cb.xy, cb.zw}, cy[4] = {ya.xy, ya.zw, yb.xy, yb.zw}, rr[4] = {ra.xy, ra.zw, rb.xy, rb.zw};
        unsigned m = 0u;
#pragma unroll
        for (int k = 3; k >= 0; --k) {                      // last circle first: alignbit pushes earlier ones to higher bits
            const vdyn_f2 ex = cx[k] - qx, ey = cy[k] - qy, reach = rr[k] + u2;
            vdyn_f2 sgn = __builtin_elementwise_fma(-ex, ex, reach * reach);
            sgn = __builtin_elementwise_fma(-ey, ey, sgn);
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(sgn.y), 31);
            m = __builtin_amdgcn_alignbit(m, __float_as_uint(sgn.x), 31);
        }
        return m;     // circle b0 + j at bit j (j = 7 was pushed first)
    }
};
template <> struct Circles8<double> {
    double cx[8], cy[8], rr[8];
    __device__ __forceinline__ void load(const double *cxr, int o, int b0)
    {
#pragma unroll
        for (int j = 0; j < 8; ++j) { cx[j] = cxr[b0 + j]; cy[j] = cxr[b0 + j + o]; rr[j] = cxr[b0 + j + 2 * o]; }
    }
    __device__ __forceinline__ unsigned skip(double x, double y, double U) const
    {
        unsigned m = 0u;
#pragma unroll
        for (int j = 7; j >= 0; --j) {
            const double ex = cx[j] - x, ey = cy[j] - y, reach = rr[j] + U;
            const double sgn = ::fma(-ey, ey, ::fma(-ex, ex, reach * reach));
            m = (m << 1) | (unsigned)(__double2hiint(sgn) >> 31 & 1);
        }
        return m;
    }
};
template <typename T>
__device__ __forceinline__ unsigned circle_skip8(const T *cxr, int o, int b0, T x, T y, T U)
{
    Circles8<T> c;
    c.load(cxr, o, b0);
    return c.skip(x, y, U);
}

// The scan of nearest_in_range<T, false> over the sub-blocks named by `mask` (bit j = sub-block sb0 + j, visited in
// increasing order: first minimum wins), eight waypoints per trip; the next sub-block's rows are read behind the
// current one's arithmetic.  `ambiguous`: some OTHER scanned waypoint lies within the tie band of the minimum -- a
// superset of what the sequential scan flags (a candidate close to the then-running minimum), so the caller's exact
// re-scan settles every case the reference's rounded roots could decide differently.
template <typename T, typename WP>
__device__ __forceinline__ void nearest_in_subblocks(const WP &wp, int sb0, unsigned mask, T x, T y, T &best_d2,
                                                     int &best_i, bool &ambiguous)
{
    using L = Lib<T>;
    const T inf = T(INFINITY);
    T best = inf, second = inf;
    int best_sb = -1;
    T dw[8];                                                          // the eight distances of the sub-block that holds `best`
#pragma unroll
    for (int k = 0; k < 8; ++k) dw[k] = inf;
    // two sub-blocks per trip (their sixteen distances are independent work for a lone wave; half the loop overhead)
    while (__any(mask != 0u)) {
        const bool act0 = mask != 0u;
        const int sba = sb0 + (act0 ? __builtin_ctz(mask) : 0);      // a finished lane re-reads its first sub-block
        mask &= mask - 1u;
        const bool act1 = mask != 0u;
        const int sbb = act1 ? sb0 + __builtin_ctz(mask) : sba;
        mask &= mask - 1u;
        Block8<T> A, B;
        A.load(wp.base, wp.yo, sba * kSubBlock);
        B.load(wp.base, wp.yo, sbb * kSubBlock);
        T da[8], db[8];
        A.d2(x, y, da);
        B.d2(x, y, db);
        T ma = min_t(min_t(min_t(da[0], da[1]), min_t(da[2], da[3])), min_t(min_t(da[4], da[5]), min_t(da[6], da[7])));
        T mb = min_t(min_t(min_t(db[0], db[1]), min_t(db[2], db[3])), min_t(min_t(db[4], db[5]), min_t(db[6], db[7])));
        ma = act0 ? ma : inf;
        mb = act1 ? mb : inf;
        // the pair's minimum, the earlier sub-block on ties; the loser is a candidate for `second`
        const bool b_wins = mb < ma;
        const T m = b_wins ? mb : ma, other = b_wins ? ma : mb;
        const int sbm = b_wins ? sbb : sba;
        const bool better = m < best;
        second = min_t(better ? best : min_t(second, m), other);
        best = better ? m : best;
        best_sb = better ? sbm : best_sb;
        // the winner's eight distances stay in registers (two selects each per trip; a lane makes one or two trips):
        // the resolve below used to re-read the winning sub-block and recompute them -- four LDS reads, thirty-two
        // instructions and a round trip on the update's critical path
#pragma unroll
        for (int k = 0; k < 8; ++k) dw[k] = better ? (b_wins ? db[k] : da[k]) : dw[k];
    }
    // the winner inside its sub-block: the first of the eight that equals the minimum
    const int sbr = best_sb < 0 ? sb0 : best_sb;
    const T thr = best * (T(2) - L::kTieBand);                        // 1 + 16 ulp
    int k_first = 0, close = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        k_first = dw[k] == best ? k : k_first;
        close += dw[k] <= thr ? 1 : 0;
    }
    best_d2 = best;
    best_i = best_sb < 0 ? 0 : sbr * kSubBlock + k_first;            // nothing comparable: the scan's initial 0
    ambiguous = best < inf && (close > 1 || second <= thr);
}

// The exact form of the same scan, for the waves in which some lane's candidates tied (1.4 % of the searches of the
// bench workload): the reference compares ROUNDED ROOTS with a strict '<' (stanley_controller.py:60-66), so the
// answer is the first waypoint whose rounded distance equals the smallest rounded distance -- taken here over the
// masked sub-blocks only (everything else is farther than the bound, no tie possible), eight IEEE square roots per
// trip.  ~1000 cycles.  (Round 2 sent such a wave back through the whole pruned search in its sequential exact form,
// ≈10 000 cycles; with one wave per SIMD the kernel ends with its slowest wave, and the unluckiest of 1024 waves
// met three or four ties in 20 updates: 35 us between the average wave's end and the kernel's.)
template <typename T, typename WP>
__device__ __forceinline__ void nearest_in_subblocks_exact(const WP &wp, int sb0, unsigned mask, T x, T y, T &best_d2,
                                                           int &best_i)
{
    using L = Lib<T>;
    const T inf = T(INFINITY);
    T best = inf;                                                     // smallest rounded root so far
    int best_sb = -1;
    while (__any(mask != 0u)) {
        const bool act = mask != 0u;
        const int sb = sb0 + (act ? __builtin_ctz(mask) : 0);
        mask &= mask - 1u;
        Block8<T> A;
        A.load(wp.base, wp.yo, sb * kSubBlock);
        T d2[8], r[8];
        A.d2(x, y, d2);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = L::sqrt(d2[k]);
        T m = min_t(min_t(min_t(r[0], r[1]), min_t(r[2], r[3])), min_t(min_t(r[4], r[5]), min_t(r[6], r[7])));
        m = act ? m : inf;
        const bool better = m < best;                                 // strict: an equal root in a later sub-block loses
        best = better ? m : best;
        best_sb = better ? sb : best_sb;
    }
    const int sbr = best_sb < 0 ? sb0 : best_sb;
    Block8<T> win;
    win.load(wp.base, wp.yo, sbr * kSubBlock);
    T d2[8];
    win.d2(x, y, d2);
    int k_first = 0;
    T d2_first = inf;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        const bool hit = L::sqrt(d2[k]) == best;
        k_first = hit ? k : k_first;
        d2_first = hit ? d2[k] : d2_first;
    }
    best_d2 = best_sb < 0 ? inf : d2_first;
    best_i = best_sb < 0 ? 0 : sbr * kSubBlock + k_first;
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
// the result is the global first minimum either way.  `adv`: how far the nearest index moved between the last two
// updates; the bound is the smallest distance to four waypoints spread over hint .. hint + 1.5 adv (where the
// vehicle is expected now) -- still distances to actual waypoints, so still an upper bound of the minimum.
// LDS image (WP::kSoa): a second level of circles, one per 8 waypoints, under the 32-waypoint ones.  Of the
// sub-blocks of blocks lo..hi a lane keeps, as bits of a mask, those whose circle reaches within U, and visits only
// them (nearest_in_subblocks): typically 2-4 sub-blocks = 16-32 waypoints instead of 64-96, and the wave's trip count
// is the largest popcount among its lanes, not the longest range.
template <typename T, bool EXACT, typename WP>
__device__ __forceinline__ void nearest_waypoint_pruned(const WP &wp, T x, T y, T &best_d2, int &best_i,
                                                        bool &ambiguous, int hint = -1, int adv = 0)
{
    best_d2 = T(INFINITY);
    best_i = 0;
    ambiguous = false;
    const int nb = (wp.W + kWpBlock - 1) / kWpBlock;
    constexpr int kChunk = 8;
    T U = T(INFINITY);
    PhaseClock pc;
    VDYN_PHASE_START(pc);
    // LDS image: the first sixteen 32-waypoint circles are on their way before the bound they will be tested against
    // is known (their reads do not depend on it; one LDS round trip less on the update's critical path)
    Circles8<T> first0, first1;
    if constexpr (WP::kSoa && !EXACT) {
        first0.load(wp.bounds, wp.bo, 0);
        first1.load(wp.bounds, wp.bo, kChunk);
    }
    const bool hinted = __all(hint >= 0) != 0;                   // wave-uniform: every lane brought a hint
    if (hinted) {
        const int last = wp.W - 1;
        T u2 = T(INFINITY);
#pragma unroll
        for (int k = 0; k < (WP::kSoa ? 4 : 1); ++k) {
            T hx, hy;
            wp.get(min(max(hint + ((k * adv) >> 1), 0), last), hx, hy);
            const T ex = hx - x, ey = hy - y;
            const T e2 = ex * ex + ey * ey;
            u2 = e2 < u2 ? e2 : u2;                               // NaN never lowers the bound
        }
        U = (T)__builtin_amdgcn_sqrtf((float)u2) * T(1.000001) + T(1e-18);
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
    VDYN_PHASE_LAP(pc, 0);      // bound U
    int lo = nb, hi = -1;
    if constexpr (WP::kSoa && !EXACT) {
        // LDS image: both levels of circles through circle_skip8, the survivors as bits of a mask
        const bool bounded = U < T(INFINITY) && x == x && y == y;        // no bound / NaN query: everything is a candidate
        for (int g0 = 0; g0 < wp.nbu; g0 += 32) {
            unsigned skip = 0u;
            for (int c = 0; c < min(32, wp.nbu - g0); c += 2 * kChunk) { // two chunks per trip: twelve reads in flight
                if (g0 + c == 0)
                    skip |= first0.skip(x, y, U) | first1.skip(x, y, U) << kChunk;
                else
                    skip |= (circle_skip8(wp.bounds, wp.bo, g0 + c, x, y, U) |
                             circle_skip8(wp.bounds, wp.bo, g0 + c + kChunk, x, y, U) << kChunk) << c;
            }
            unsigned need = bounded ? ~skip : ~0u;
            const int valid = nb - g0;                                    // blocks of THIS lane's table in the group
            need &= valid >= 32 ? ~0u : ((1u << max(valid, 0)) - 1u);
            lo = need != 0u ? min(lo, g0 + __builtin_ctz(need)) : lo;
            hi = need != 0u ? max(hi, g0 + 31 - __builtin_clz(need)) : hi;
        }
        if (hi < lo) { lo = 0; hi = nb - 1; }                             // nothing comparable: everything
        VDYN_PHASE_LAP(pc, 1);      // 32-waypoint circles
        const int nsb = (wp.W + kSubBlock - 1) / kSubBlock;
        const int sb0 = lo * kSubPerBlock;
        const int nsub = min((hi + 1) * kSubPerBlock, nsb) - sb0;        // sub-blocks of the lane's range, >= 1
        if (__all(nsub <= 32)) {                                          // (longer ranges: the plain scan below)
            unsigned skip = 0u;
            for (int j0 = 0; __any(j0 < nsub); j0 += 2 * kChunk) {        // a lane past its range reads the padding
                const int j = min(sb0 + j0, wp.nsbu);
                skip |= (circle_skip8(wp.sub, wp.so, j, x, y, U) | circle_skip8(wp.sub, wp.so, j + kChunk, x, y, U) << kChunk) << j0;
            }
            unsigned mask = bounded ? ~skip : ~0u;
            mask &= nsub >= 32 ? ~0u : ((1u << nsub) - 1u);
            VDYN_PHASE_LAP(pc, 2);      // 8-waypoint circles
            nearest_in_subblocks<T>(wp, sb0, mask, x, y, best_d2, best_i, ambiguous);
            if (__builtin_expect(__any(ambiguous) != 0, 0)) {          // wave-uniform; settles every lane of the wave
                VDYN_EVENT(1);
                nearest_in_subblocks_exact<T>(wp, sb0, mask, x, y, best_d2, best_i);
                ambiguous = false;
            }
            VDYN_PHASE_LAP(pc, 3);      // block scan + resolve
            return;
        }
    } else {
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
    }
    nearest_in_range<T, EXACT>(wp, lo * kWpBlock, min((hi + 1) * kWpBlock, wp.W), x, y, best_d2, best_i, ambiguous);
    if constexpr (!EXACT) {
        // a tie: the exact form of the scan over the SAME range (what lies outside it is farther than the bound and
        // cannot tie) -- not, as in round 2, the whole search again from the circles on
        if (__builtin_expect(__any(ambiguous) != 0, 0)) {
            VDYN_EVENT(2);
            best_d2 = T(INFINITY);
            best_i = 0;
            nearest_in_range<T, true>(wp, lo * kWpBlock, min((hi + 1) * kWpBlock, wp.W), x, y, best_d2, best_i, ambiguous);
            ambiguous = false;
        }
    }
}

// stanley_controller.py:78-129 -> steering angle (limited), target index, crosstrack error
template <typename T, typename WP>
__device__ __forceinline__ void stanley_control(const CtrlGains<T> &G, const WP &wp, T x, T y, T yaw,
                                                T v, T &steer_out, int &idx_out, T &cte_out, int *near_io = nullptr,
                                                int *adv_io = nullptr)
{
    using L = Lib<T>;
    T best_d2;
    int best_i;
    bool amb;
    if (wp.bounds != nullptr) {
        const int hint = near_io != nullptr ? *near_io : -1;
        const int adv = adv_io != nullptr ? *adv_io : 0;
        nearest_waypoint_pruned<T, false>(wp, x, y, best_d2, best_i, amb, hint, adv);
        VDYN_EVENT(0);
        if (__builtin_expect(__any(amb) != 0, 0)) {      // never taken any more: the pruned search settles its own ties
            if (amb) nearest_waypoint_pruned<T, true>(wp, x, y, best_d2, best_i, amb, hint, adv);
        }
        if (near_io != nullptr) {
            if (adv_io != nullptr) *adv_io = hint >= 0 ? best_i - hint : 0;
            *near_io = best_i;
        }
    } else {
        nearest_waypoint<T, false>(wp, x, y, best_d2, best_i, amb);
        if (__builtin_expect(__any(amb) != 0, 0)) {
            if (amb) nearest_waypoint<T, true>(wp, x, y, best_d2, best_i, amb);
        }
    }
    PhaseClock pc;
    VDYN_PHASE_START(pc);
    // :68-76 walk forward until the accumulated arc length reaches the lookahead distance
    T total = L::sqrt_fast(best_d2);
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
            int lo = best_i, hi = last;                          // invariant: cum[lo] < target; answer in (lo, hi]
            // waypoints per metre (inf / NaN: the bisection decides); the LDS image keeps it per lane, so that the
            // window below is read together with cum[best_i] instead of after cum[last]
            const T per = WP::kSoa ? wp.per : (T)last / wp.seg_at(last);
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
            // Window entries below the target raise lo, entries at or above it lower hi.  cum is a running sum of
            // non-negative lengths: non-decreasing, and a NaN (a non-finite waypoint) poisons everything after it -- so
            // along the window the entries are a run of "below", then a run of "at or above", then NaNs, and two
            // COUNTS say where the runs meet (two instructions per entry and count, where a compare-and-select per
            // entry and bound took nineteen: a sixth of the update's instructions went into this loop).
            int n_below = 0, n_above = 0;
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                n_below += cw[k] < target ? 1 : 0;
                n_above += cw[k] >= target ? 1 : 0;          // false for NaN: never lowers hi
            }
            lo = n_below > 0 ? max(lo, min(w0 + n_below - 1, last)) : lo;
            hi = n_above > 0 ? min(hi, min(w0 + n_below, last)) : hi;
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
        // A non-finite waypoint poisons the cumulative arc from there to the table's end: the search above then
        // answers best_i + 1 (every comparison with a NaN fails), where the reference -- which adds the segments one by
        // one from the nearest waypoint (:68-76) -- walks on normally when the bad row lies BEHIND the vehicle and runs
        // to the last waypoint when the walk crosses it (a NaN total is never >= the lookahead).  Such a table is
        // known by its end, cum[last], which the lane holds as `per`: its vehicles take the reference's walk, on the
        // coordinates themselves.  No lane does on a finite table: one compare and a branch never taken.
        const T per_ = WP::kSoa ? wp.per : (T)last / wp.seg_at(last);
        const bool poisoned = !(per_ == per_) || per_ == T(0);       // cum[last] NaN / inf
        if (__builtin_expect(__any(poisoned) != 0, 0)) {
            if (poisoned) {
                T tot = total, ax_, ay_;
                int c2 = best_i;
                wp.get(best_i, ax_, ay_);
                for (int i = best_i + 1; i < wp.W; ++i) {
                    if (tot >= G.lookahead) break;
                    T qx, qy;
                    wp.get(i, qx, qy);
                    tot += segment_length<T>(ax_, ay_, qx, qy);
                    c2 = i;
                    ax_ = qx;
                    ay_ = qy;
                }
                ce = c2;
            }
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
    VDYN_PHASE_LAP(pc, 4);      // lookahead
    // :90-98 (px, py) is waypoint ce
    T sy, cy;
    L::sincos(yaw, &sy, &cy);
    const T v0 = px - x - G.lookahead * cy;
    const T v1 = py - y - G.lookahead * sy;
    T cte = L::sqrt_fast(v0 * v0 + v1 * v1);
    if (cte < G.deadband) cte = T(0);
    // :101-104: only the SIGN of the wrapped angle between the vector to the target and the heading is used.  fp64
    // forms the angle as the reference does; fp32 takes the sign of sin(angle) |v| = v1 cos(yaw) - v0 sin(yaw), which
    // is the same sign on (-pi, pi) without an arctangent and a wrap (~40 instructions of an update's ~1200), and at
    // least as well conditioned as the difference of two rounded angles
    T che;
    if constexpr (sizeof(T) == 4) che = v1 * cy - v0 * sy;
    else che = wrap_pi<T>(L::atan2(v1, v0) - yaw);
    const T sign = che > T(0) ? T(1) : (che < T(0) ? T(-1) : che);  // np.sign: 0 and nan pass through
    // :109-120 trajectory heading; wraps from the last waypoint to the first
    T ax_, ay_, bx_, by_;
    {   // both ends by index, without a branch: the two reads go out together (and with the one of waypoint ce)
        const bool inner = ce < wp.W - 1;
        wp.get(inner ? ce : wp.W - 1, ax_, ay_);
        wp.get(inner ? ce + 1 : 0, bx_, by_);
    }
    T he;                                                                  // :122-123
    if constexpr (sizeof(T) == 4) {
        // the segment direction turned into the vehicle's frame first: atan2 of (cross, dot) IS the wrapped difference.
        // A zero-length segment (a closed path whose last waypoint repeats the first, a one-waypoint table, a
        // duplicated waypoint) has the reference's arctan2(0, 0) = 0 as its heading (:109-120), i.e. the wrapped
        // -yaw = atan2(-sin yaw, cos yaw): (cross, dot) of the unit vector along +x, not the atan2(+-0, +-0) of the
        // rotated zero vector (0 or +-pi by the signs of sin / cos yaw)
        const T dx_ = bx_ - ax_, dy_ = by_ - ay_;
        const bool degenerate = dx_ == T(0) && dy_ == T(0);
        const T cross = degenerate ? -sy : dy_ * cy - dx_ * sy;
        const T dot = degenerate ? cy : dx_ * cy + dy_ * sy;
        he = L::atan2(cross, dot);
    } else {
        he = wrap_pi<T>(L::atan2(by_ - ay_, bx_ - ax_) - yaw);
    }
    T steer = he + L::atan(L::div(G.k * sign * cte, v + G.k_soft));        // :124-126
    steer = steer < -G.max_steer ? -G.max_steer : steer;                   // :128 np.clip
    steer = steer > G.max_steer ? G.max_steer : steer;
    VDYN_PHASE_LAP(pc, 5);      // steering law
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
    const T d = Lib<T>::div(G.kd * (current - prev), dt);       // fp64: the division; fp32: times v_rcp_f32(dt)
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
    int adv = 0;    // how far `near` moved between the last two updates (where to expect the vehicle now; prunes only)
};

// One controller update, drive.py:128-138, from the current vehicle state s[10];
// `steer` returns the unfiltered (limited) Stanley angle.
template <typename T, typename WP>
__device__ __forceinline__ void controller_update(const CtrlGains<T> &G, const WP &wp, const T s[10],
                                                  T dt, CtrlState<T> &c, T &steer)
{
    PhaseClock pc;
    VDYN_PHASE_START(pc);
    stanley_control<T>(G, wp, s[8], s[9], s[7], s[0], steer, c.idx, c.cte, &c.near, &c.adv);  // drive.py:129-130
    VDYN_PHASE_LAP(pc, 8);      // all of stanley_control
    long_control<T>(G, c.target, s[0], c.prev_vel, c.total, dt, c.tau);       // :131-133
    c.prev_vel = s[0];                                                        // :134
    c.x_del = G.filt_keep * c.x_del + G.filt_gain * steer;                    // :137
    c.delta = c.x_del;                                                        // :138
    VDYN_PHASE_LAP(pc, 9);      // PID + filter
}

}  // namespace vdyn
