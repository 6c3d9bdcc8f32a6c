#!/usr/bin/env python3
"""Derive the minimax polynomial coefficients used by csrc/vdyn_fastmath.hpp.

Lawson-weighted least squares in float64 on a dense grid, coefficients then
rounded to float32 and the fp32 Horner evaluation checked against float64 libm.
Run:  python3 tools/fit_polys.py
"""
import numpy as np


def lawson(u, target, deg, weight=None, iters=60):
    """min max |weight * (P(u) - target)| over polynomials of degree `deg` in u."""
    w = np.ones_like(u)
    wt = np.ones_like(u) if weight is None else weight
    V = np.polynomial.chebyshev.chebvander(2 * (u - u.min()) / (u.max() - u.min()) - 1, deg)
    for _ in range(iters):
        A = V * (w * wt)[:, None]
        c, *_ = np.linalg.lstsq(A, target * w * wt, rcond=None)
        err = np.abs((V @ c - target) * wt)
        w = w * (err / err.max() + 1e-3)
        w /= w.sum()
    # convert Chebyshev-on-[a,b] to monomials in u
    cheb = np.polynomial.chebyshev.Chebyshev(c, domain=[u.min(), u.max()])
    return cheb.convert(kind=np.polynomial.Polynomial).coef, err.max()


def horner32(coef, u):
    acc = np.full_like(u, np.float32(coef[-1]), dtype=np.float32)
    for c in coef[-2::-1]:
        acc = (acc * u + np.float32(c)).astype(np.float32)
    return acc


def report(name, coef, err):
    print(f"// {name}: max fit error {err:.3e}")
    print("//   " + ", ".join(f"{np.float32(c):.9e}f" for c in coef))


if __name__ == "__main__":
    # atan(t) = t * P(t^2), t in [0, 1]  (relative error)
    t = np.linspace(1e-6, 1.0, 200001)
    for deg in (6, 7, 8):
        coef, err = lawson(t * t, np.arctan(t) / t, deg)
        report(f"atan(t)/t in u=t^2, degree {deg}", coef, err)
        t32 = t.astype(np.float32)
        got = (t32 * horner32(coef, (t32 * t32).astype(np.float32))).astype(np.float64)
        print(f"//   fp32 eval max rel err {np.max(np.abs(got - np.arctan(t32.astype(np.float64))) / np.arctan(t32.astype(np.float64))):.3e}")

    # sin(r) = r + r^3 * S(r^2), r in [-pi/2, pi/2]   (relative error)
    r = np.linspace(1e-6, np.pi / 2 + 1e-3, 200001)
    for deg in (3, 4):
        coef, err = lawson(r * r, (np.sin(r) - r) / r ** 3, deg, weight=r * r)
        report(f"(sin r - r)/r^3 in u=r^2, degree {deg}, |r|<=pi/2", coef, err)
        r32 = r.astype(np.float32)
        u = (r32 * r32).astype(np.float32)
        got = (r32 + (r32 * u * horner32(coef, u)).astype(np.float32)).astype(np.float64)
        print(f"//   fp32 eval max rel err {np.max(np.abs(got - np.sin(r32.astype(np.float64))) / np.sin(r32.astype(np.float64))):.3e}")

    # sin / cos kernels on [-pi/4, pi/4]
    r = np.linspace(1e-6, np.pi / 4 + 1e-3, 200001)
    for deg in (2, 3):
        coef, err = lawson(r * r, (np.sin(r) - r) / r ** 3, deg, weight=r * r)
        report(f"(sin r - r)/r^3, degree {deg}, |r|<=pi/4", coef, err)
        coef, err = lawson(r * r, (np.cos(r) - 1 + r * r / 2) / r ** 4, deg, weight=r ** 4)
        report(f"(cos r - 1 + r^2/2)/r^4, degree {deg}, |r|<=pi/4", coef, err)
