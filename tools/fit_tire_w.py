#!/usr/bin/env python3
"""The fit behind the FAST step's tire chain (csrc/vdyn_kernels.hip, fit_tire_wheel):

    sin(C atan x) / x = c W_C(c),   c = 1 / sqrt(1 + x^2),   W_C(c) = sin(C acos c) / sqrt(1 - c^2)

W_C -- U_{C-1}, the Chebyshev polynomial of the second kind, continued to non-integer C -- is analytic on
(-1, 1]; on c in [0, 1] (every slip from infinity down to 0) its Chebyshev series converges like
(3 + sqrt 8)^-n = 5.83^-n.  This script reproduces the library's fit in NumPy (interpolation at the Chebyshev
nodes of [0, 1], monomials in c, rounded to the working precision) and prints, per degree, the error of the Horner
evaluation in that precision against float64 / long double:

    python3 tools/fit_tire_w.py            # table for C = 1.5047 (the reference's), 1.3, 1.9, 2.0

fp32 (one rounding per step, as the packed fma does): degree 7 5.5e-7, **degree 8 2.2e-7**, degree 9+ no better
(rounding-limited).  fp64: **degree 16 3e-14** (the library's since round 4: two fmas per wheel and stage less), degree 18
2e-15 (rounds 2-3), degree 20+ no better; output committed as profiles/r04_tire_fit_degrees.txt.  The library's own
coefficients for a given C come from `VehicleModel.tire_fit(C)` (`vdyn_tire_fit_f32 / _f64`)."""
import numpy as np
from numpy.polynomial import chebyshev as Ch


def w_exact(c, C):
    c = np.asarray(c, dtype=np.longdouble)
    th = np.arccos(np.clip(c, -1, 1))
    small = th < 1e-5
    return np.where(small, C * (1 + (1 - C * C) * th * th / 6), np.sin(C * th) / np.where(small, 1, np.sin(th)))


def fit(C, deg):
    k = np.arange(deg + 1)
    z = np.cos(np.pi * (2 * k + 1) / (2 * (deg + 1)))
    co = Ch.cheb2poly(Ch.chebfit(z, w_exact((z + 1) / 2, C).astype(np.float64), deg))     # monomials in z = 2 c - 1
    return np.polynomial.Polynomial(co)(np.polynomial.Polynomial([-1, 2])).coef[::-1]  # in c, highest degree first


def errors(C, deg, dtype):
    co = fit(C, deg).astype(dtype)
    x = np.concatenate([np.linspace(0, 4, 200001), np.geomspace(1e-8, 1e8, 100001)])
    c = (1 / np.sqrt((1 + (x * x).astype(dtype)).astype(np.float64))).astype(dtype)
    g = np.full_like(c, co[0])
    for a in co[1:]:
        g = (g.astype(np.float64) * c.astype(np.float64) + np.float64(a)).astype(dtype) if dtype == np.float32 else g * c + a
    G = (g.astype(np.float64) * c.astype(np.float64)).astype(dtype).astype(np.longdouble)
    xl = x.astype(np.longdouble)
    want = np.where(xl > 0, np.sin(C * np.arctan(xl)) / np.where(xl > 0, xl, 1), C)
    small = x <= np.sqrt(3.0)
    return float(np.max(np.abs(G - want) * xl)), float(np.max((np.abs(G - want) / np.abs(want))[small]))


def by_c():
    """The library's degrees (fp32: 8, fp64: 16) over the whole range of shape factors a handle's fit is accepted for
    (0 .. 2.9; gates 5e-7 / 5e-14, csrc/vdyn_kernels.hip kTireFitTol*): python3 tools/fit_tire_w.py --by-c
    (committed as profiles/r05_tire_fit_by_C.txt; ADVICE round 4: the margin per C, not only at the reference's C)."""
    cs = np.round(np.arange(0.05, 2.951, 0.05), 2)
    for dtype, deg, gate in ((np.float32, 8, 5e-7), (np.float64, 16, 5e-14)):
        rows = [(C,) + errors(float(C), deg, dtype) for C in cs]
        worst = max(rows, key=lambda r: max(r[1], r[2]))
        print(f"{np.dtype(dtype).name}, degree {deg}, gate {gate:g}: C  |error of sin(C atan x)|  relative error of sin(C atan x)/x (x <= sqrt 3)")
        for C, a, r in rows:
            print(f"  {C:4.2f}  {a:.2e}  {r:.2e}" + ("   <-- over the gate: such a handle keeps the general chain" if max(a, r) > gate else ""))
        print(f"  worst: C = {worst[0]:.2f}: {max(worst[1], worst[2]):.2e} = {max(worst[1], worst[2]) / gate:.2f} of the gate; "
              f"the reference's C = 1.5047: {max(errors(1.5047, deg, dtype)):.2e}")


if __name__ == "__main__":
    import sys
    if "--by-c" in sys.argv:
        by_c()
        sys.exit(0)
    for dtype, degs in ((np.float32, (5, 6, 7, 8, 9, 10)), (np.float64, (12, 14, 16, 18, 20, 22))):
        print(f"{np.dtype(dtype).name}: max |error of sin(C atan x)| / max relative error of sin(C atan x)/x for x <= sqrt(3)")
        for C in (1.5047, 1.3, 1.9, 2.0):
            print(f"  C = {C:<7}" + "  ".join(f"deg {d}: {a:.1e}/{r:.1e}" for d in degs for a, r in [errors(C, d, dtype)]))
