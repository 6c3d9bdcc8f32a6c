// vdyn_lattice.hpp -- lattice generation per lane, for gfx950.
//
// Semantics:
//   /root/reference/libs/motionplanner/local_planner.py:25-52    get_closest_index
//   /root/reference/libs/motionplanner/local_planner.py:85-152   get_goal_index
//   /root/reference/libs/motionplanner/local_planner.py:154-275  get_goal_state_set
//   /root/reference/libs/motionplanner/path_optimizer.py:31-88   optimize_spiral
//   /root/reference/libs/motionplanner/path_optimizer.py:131-175 sample_spiral
//   /root/reference/libs/motionplanner/local_planner.py:317-321  path validity
//   /root/reference/libs/motionplanner/local_planner.py:424-470  transform_paths
//   /root/reference/libs/motionplanner/local_planner.py:395-419  waypoint re-interpolation
//
// Everything but the optimiser is restated exactly.  The reference minimises its objective
// (path_optimizer.py:183-198: bending energy + 25 (x_f, y_f errors)^2 + 30 (theta_f error)^2,
// end point by an 8-interval Simpson rule) with SciPy's L-BFGS-B, whose iterates live in
// Fortran outside the reference.  The objective is a sum of five squares in (p1, p2, sf), so the
// device minimises the SAME objective over the SAME box with a projected Levenberg-Marquardt
// iteration on an analytic Jacobian: same minimiser, reached to a tighter tolerance than
// L-BFGS-B's stopping rule (tests compare the objective values and the resulting paths).
#pragma once
#include <hip/hip_runtime.h>

#include "vdyn_controls.hpp"

namespace vdyn {

constexpr int kSpiralPoints = 49;  // x, y entries of a sampled spiral (np.linspace's 50 minus the origin)

// ---- local_planner.py:25-52: '<=' on the sqrt distances -> the LAST of equal minima wins ----
// Same structure as the Stanley scan (vdyn_controls.hpp): squared distances in an unrolled,
// branch-free loop; a sticky flag marks a candidate a few ulp ABOVE the running minimum (its
// rounded root may still tie, and the reference would then move to the later index); flagged
// lanes repeat the scan comparing the roots themselves.
template <typename T, bool EXACT>
__device__ __forceinline__ void closest_scan(const T *__restrict__ px, const T *__restrict__ py, int n, T ex, T ey,
                                             T &best_d2, int &best_i, bool &ambiguous)
{
    using L = Lib<T>;
    best_d2 = T(INFINITY);
    best_i = 0;
    ambiguous = false;
    const T band = T(2) - L::kTieBand;  // 1 + 16 ulp
#pragma unroll 8
    for (int i = 0; i < n; ++i) {
        const T dx = px[i] - ex, dy = py[i] - ey;
        const T d2 = dx * dx + dy * dy;
        bool take = d2 <= best_d2;
        const bool close = !take && d2 <= best_d2 * band;
        if (EXACT) {
            if (close) take = L::sqrt(d2) <= L::sqrt(best_d2);
        } else {
            ambiguous = ambiguous || close;
        }
        best_d2 = take ? d2 : best_d2;
        best_i = take ? i : best_i;
    }
}

template <typename T>
__device__ __forceinline__ void closest_index(const T *__restrict__ px, const T *__restrict__ py, int n, T ex,
                                              T ey, int &idx, T &len)
{
    T best_d2;
    bool amb;
    closest_scan<T, false>(px, py, n, ex, ey, best_d2, idx, amb);
    if (__builtin_expect(__any(amb) != 0, 0)) {
        if (amb) closest_scan<T, true>(px, py, n, ex, ey, best_d2, idx, amb);
    }
    len = Lib<T>::sqrt(best_d2);
}

// ---- local_planner.py:85-152 ----
template <typename T>
__device__ __forceinline__ int goal_index(const T *__restrict__ px, const T *__restrict__ py, int n, T lookahead,
                                          T closest_len, int closest_idx)
{
    T arc = closest_len;
    int w = closest_idx;
    if (arc > lookahead || w == n - 1) return w;
    while (w < n - 1) {
        arc += segment_length<T>(px[w + 1], py[w + 1], px[w], py[w]);
        if (arc > lookahead) break;
        ++w;
    }
    return w;
}

// ---- local_planner.py:154-275: goal k of P in the vehicle frame -> (x, y, t) ----
template <typename T>
__device__ __forceinline__ void goal_state(const T *__restrict__ px, const T *__restrict__ py, int n, int gi, T ex,
                                           T ey, T eyaw, int k, int P, T path_offset, T &gx, T &gy, T &gt)
{
    using L = Lib<T>;
    const T pi = T(3.141592653589793);
    T dx, dy;
    if (gi < n - 1) { dx = px[gi + 1] - px[gi]; dy = py[gi + 1] - py[gi]; }
    else { dx = px[gi] - px[gi - 1]; dy = py[gi] - py[gi - 1]; }
    const T heading = L::atan2(dy, dx);
    const T lx = px[gi] - ex, ly = py[gi] - ey;
    T s, c;
    L::sincos(-eyaw, &s, &c);
    const T goal_x = lx * c - ly * s;
    const T goal_y = lx * s + ly * c;
    T goal_t = heading - eyaw;
    if (goal_t > pi) goal_t -= 2 * pi;
    else if (goal_t < -pi) goal_t += 2 * pi;
    const T offset = (T)(k - P / 2) * path_offset;
    T so, co;
    L::sincos(goal_t + pi / 2, &so, &co);
    gx = goal_x + offset * co;
    gy = goal_y + offset * so;
    gt = goal_t;
}

// ---- the spiral objective as five residuals and their Jacobian -----------------------------
// theta(u sf) = sf (A(u) p1 + B(u) p2) for the cubic spiral with zero end curvatures
// (path_optimizer.py:149-154 with p0 = p3 = 0).
template <typename T>
__device__ __forceinline__ void spiral_basis(T u, T &A, T &B)
{
    const T u2 = u * u;
    A = u2 * (T(4.5) + u * (T(-7.5) + T(3.375) * u));
    B = u2 * (T(-2.25) + u * (T(6.0) - T(3.375) * u));
}

// r[5], Jac[5][3] at (p1, p2, sf); J = sum r^2 equals path_optimizer.py:183-189
template <typename T>
__device__ __forceinline__ T spiral_residuals(T p1, T p2, T sf, T xf, T yf, T tf, T r[5], T Jc[5][3])
{
    using L = Lib<T>;
    T xs = 0, ys = 0, dx1 = 0, dx2 = 0, dx3 = 0, dy1 = 0, dy2 = 0, dy3 = 0;
#pragma unroll
    for (int i = 0; i <= 8; ++i) {
        const T w = (i == 0 || i == 8) ? T(1) : ((i & 1) ? T(4) : T(2));   // Simpson
        T A, B, s, c;
        spiral_basis<T>((T)i * T(0.125), A, B);
        const T g = A * p1 + B * p2;
        L::sincos(sf * g, &s, &c);
        xs += w * c; ys += w * s;
        dx1 -= w * s * A; dx2 -= w * s * B; dx3 -= w * s * g;
        dy1 += w * c * A; dy2 += w * c * B; dy3 += w * c * g;
    }
    const T k24 = T(1.0 / 24.0), sq30 = T(5.477225575051661);
    // Cholesky factor of the bending-energy form (324 p1^2 - 81 p1 p2 + 324 p2^2) / 840
    const T L11 = T(0.6210590034081187), L21 = T(-0.07763237542601484), L22 = T(0.6161878771933119);
    const T rs = L::sqrt(sf), A1 = T(0.375), B1 = T(0.375);
    r[0] = T(5) * (xf - sf * xs * k24);
    r[1] = T(5) * (yf - sf * ys * k24);
    r[2] = sq30 * (tf - sf * (A1 * p1 + B1 * p2));
    r[3] = rs * (L11 * p1 + L21 * p2);
    r[4] = rs * (L22 * p2);
    const T k = sf * sf * k24;
    Jc[0][0] = T(-5) * k * dx1; Jc[0][1] = T(-5) * k * dx2; Jc[0][2] = T(-5) * k24 * (xs + sf * dx3);
    Jc[1][0] = T(-5) * k * dy1; Jc[1][1] = T(-5) * k * dy2; Jc[1][2] = T(-5) * k24 * (ys + sf * dy3);
    Jc[2][0] = -sq30 * sf * A1; Jc[2][1] = -sq30 * sf * B1; Jc[2][2] = -sq30 * (A1 * p1 + B1 * p2);
    Jc[3][0] = rs * L11; Jc[3][1] = rs * L21; Jc[3][2] = (L11 * p1 + L21 * p2) / (T(2) * rs);
    Jc[4][0] = T(0); Jc[4][1] = rs * L22; Jc[4][2] = (L22 * p2) / (T(2) * rs);
    T J = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) J += r[i] * r[i];
    return J;
}

// Minimise over p1, p2 in [-0.5, 0.5], sf >= sf0 from (0, 0, sf0) (path_optimizer.py:58-84).
template <typename T>
__device__ __forceinline__ T optimize_spiral(T xf, T yf, T tf, T p[3], int &iters)
{
    using L = Lib<T>;
    const T sf0 = L::sqrt(xf * xf + yf * yf);
    const T lo[3] = {T(-0.5), T(-0.5), sf0}, hi[3] = {T(0.5), T(0.5), T(INFINITY)};
    T x[3] = {T(0), T(0), sf0};
    T r[5], Jc[5][3];
    T J = spiral_residuals<T>(x[0], x[1], x[2], xf, yf, tf, r, Jc);
    T lambda = T(1e-3);
    const T eps = sizeof(T) == 8 ? T(1e-10) : T(1e-5);      // projected-gradient tolerance
    const T stall = sizeof(T) == 8 ? T(1e-15) : T(1e-7);    // relative decrease that counts as none
    int it = 0;
    bool done = !(sf0 > T(0)) || !(J == J);   // degenerate goal / NaN: leave the straight line
    for (int trip = 0; trip < 80 && __any(!done); ++trip) {
        it += done ? 0 : 1;
        // gradient g = J^T r and Gauss-Newton matrix H = J^T J
        T g[3] = {0, 0, 0}, H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                g[a] += Jc[i][a] * r[i];
#pragma unroll
                for (int b = 0; b < 3; ++b) H[a][b] += Jc[i][a] * Jc[i][b];
            }
        // active set: a variable sitting on a bound whose gradient pushes outward stays there
        bool fixed[3];
        T gnorm = 0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            fixed[a] = (x[a] <= lo[a] && g[a] > T(0)) || (x[a] >= hi[a] && g[a] < T(0));
            gnorm = fixed[a] ? gnorm : fmax(gnorm, abs_t(g[a]));
        }
        if (gnorm <= eps * (T(1) + J)) done = true;
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if (fixed[a]) {
                g[a] = T(0);
#pragma unroll
                for (int b = 0; b < 3; ++b) { H[a][b] = T(0); H[b][a] = T(0); }
                H[a][a] = T(1);
            }
        // (H + lambda diag(H)) d = -g by Cholesky
        const T a00 = H[0][0] * (T(1) + lambda), a11 = H[1][1] * (T(1) + lambda), a22 = H[2][2] * (T(1) + lambda);
        const T l00 = L::sqrt(a00), l10 = H[1][0] / l00, l20 = H[2][0] / l00;
        const T l11 = L::sqrt(a11 - l10 * l10), l21 = (H[2][1] - l20 * l10) / l11;
        const T l22 = L::sqrt(a22 - l20 * l20 - l21 * l21);
        const T y0 = -g[0] / l00, y1 = (-g[1] - l10 * y0) / l11, y2 = (-g[2] - l20 * y0 - l21 * y1) / l22;
        const T d2 = y2 / l22, d1 = (y1 - l21 * d2) / l11, d0 = (y0 - l10 * d1 - l20 * d2) / l00;
        T xn[3] = {x[0] + d0, x[1] + d1, x[2] + d2};
#pragma unroll
        for (int a = 0; a < 3; ++a) xn[a] = fmin(fmax(xn[a], lo[a]), hi[a]);
        T rn[5], Jn[5][3];
        const T Jnew = spiral_residuals<T>(xn[0], xn[1], xn[2], xf, yf, tf, rn, Jn);
        const bool ok = Jnew < J;                     // NaN compares false: the step is rejected
        if (!done) {
            if (ok) {
                if (J - Jnew <= stall * J) done = true;       // no further progress to be had
#pragma unroll
                for (int a = 0; a < 3; ++a) x[a] = xn[a];
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    r[i] = rn[i];
#pragma unroll
                    for (int a = 0; a < 3; ++a) Jc[i][a] = Jn[i][a];
                }
                J = Jnew;
                lambda = fmax(lambda * T(1.0 / 3.0), T(1e-12));
            } else {
                lambda *= T(4);
                if (lambda > T(1e12)) done = true;
            }
        }
    }
    p[0] = x[0]; p[1] = x[1]; p[2] = x[2];
    iters = it;
    return J;
}

// ---- path_optimizer.py:131-175 + local_planner.py:317-321,424-470 ---------------------------
// Samples the spiral, decides validity against the goal, writes the transformed path
// (rows x, y, yaw with `stride` between points).  Quirk kept: the reference pairs yaw[i] =
// theta(s_i) with x[i], y[i] = position at s_{i+1} (its index loop runs over len(x) = 49).
template <typename T>
__device__ __forceinline__ bool sample_and_transform(const T p[3], T gx, T gy, T gt, T ex, T ey, T eyaw, T *ox,
                                                     T *oy, T *ot, int64_t stride)
{
    using L = Lib<T>;
    const T p1 = p[0], p2 = p[1], sf = p[2];
    const T b = -(T(-9.0) * p1 + T(9.0) * p2 / T(2.0)) / sf;
    const T c = (T(-45.0) * p1 / T(2.0) + T(18.0) * p2) / (sf * sf);
    const T d = -(T(-27.0) * p1 / T(2.0) + T(27.0) * p2 / T(2.0)) / (sf * sf * sf);
    const T step = sf / T(49);
    T se, ce;
    L::sincos(eyaw, &se, &ce);
    T cp = T(1), sp = T(0), sx = T(0), sy = T(0), s_prev = T(0), th_prev = T(0), th_end = T(0);
    for (int i = 1; i < 50; ++i) {
        const T s = (i == 49) ? sf : (T)i * step;
        const T th = (b / 2) * (s * s) + (c / 3) * (s * s * s) + (d / 4) * ((s * s) * (s * s));
        T st, ct;
        L::sincos(th, &st, &ct);
        const T ds = s - s_prev;
        sx += ds * (ct + cp) / 2;
        sy += ds * (st + sp) / 2;
        ox[(int64_t)(i - 1) * stride] = ex + sx * ce - sy * se;
        oy[(int64_t)(i - 1) * stride] = ey + sx * se + sy * ce;
        ot[(int64_t)(i - 1) * stride] = th_prev + eyaw;
        cp = ct; sp = st; s_prev = s; th_prev = th; th_end = th;
    }
    const T e0 = sx - gx, e1 = sy - gy, e2 = th_end - gt;
    return !(L::sqrt(e0 * e0 + e1 * e1 + e2 * e2) > T(0.1));          // local_planner.py:317-321
}

}  // namespace vdyn
