#!/usr/bin/env python3
"""Double-precision polynomial coefficients for csrc/vdyn_fastmath.hpp (fm64).

Chebyshev interpolation in 60-digit arithmetic (mpmath), converted exactly to monomial
coefficients and rounded to double; the double Horner evaluation is then checked
against mpmath on a grid.  Run:  python3 tools/fit_polys_f64.py
"""
import mpmath as mp
import numpy as np

mp.mp.dps = 60


def cheb_fit(f, a, b, deg):
    """Monomial coefficients (in the variable itself) of the degree-`deg` Chebyshev
    interpolant of f on [a, b]."""
    n = deg + 1
    nodes = [mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
    xs = [(a + b) / 2 + (b - a) / 2 * t for t in nodes]
    fs = [f(x) for x in xs]
    c = []
    for j in range(n):
        s = mp.fsum(fs[k] * mp.cos(mp.pi * j * (2 * k + 1) / (2 * n)) for k in range(n)) * 2 / n
        c.append(s)
    c[0] /= 2
    # Chebyshev T_j(t) -> monomials in t
    T = [[mp.mpf(1)], [mp.mpf(0), mp.mpf(1)]]
    for j in range(2, n):
        prev, prev2 = T[-1], T[-2]
        cur = [mp.mpf(0)] + [2 * v for v in prev]
        for i, v in enumerate(prev2):
            cur[i] -= v
        T.append(cur)
    mono_t = [mp.mpf(0)] * n
    for j in range(n):
        for i, v in enumerate(T[j]):
            mono_t[i] += c[j] * v
    # t = (2x - a - b)/(b - a) = alpha x + beta
    alpha, beta = 2 / (b - a), -(a + b) / (b - a)
    mono_x = [mp.mpf(0)] * n
    for i, ci in enumerate(mono_t):      # ci * (alpha x + beta)^i
        for m in range(i + 1):
            mono_x[m] += ci * mp.binomial(i, m) * alpha ** m * beta ** (i - m)
    return mono_x


def horner64(coef, u):
    acc = np.full_like(u, float(coef[-1]))
    for c in coef[-2::-1]:
        acc = acc * u + float(c)
    return acc


def show(name, coef):
    print(f"// {name}")
    for c in coef:
        print(f"    {mp.nstr(c, 20)},")


if __name__ == "__main__":
    one = mp.mpf(1)
    # atan(t)/t as a polynomial in u = t^2 on [0, 1]
    f = lambda u: (mp.atan(mp.sqrt(u)) / mp.sqrt(u)) if u > 0 else one
    for deg in (20, 22):
        co = cheb_fit(f, mp.mpf(0), one, deg)
        t = np.linspace(1e-9, 1.0, 20001)
        got = t * horner64(co, t * t)
        want = np.array([float(mp.atan(mp.mpf(float(x)))) for x in t])
        print(f"// atan deg {deg}: max rel err {np.max(np.abs(got - want) / want):.3e}")
        if deg == 22:
            show("ATAN64: atan(t) = t * P(t^2), |t| <= 1, P degree 22", co)
    # (sin r - r)/r^3 in u = r^2 on [0, (pi/2)^2]
    lim = (mp.pi / 2 + mp.mpf("0.001")) ** 2
    g = lambda u: ((mp.sin(mp.sqrt(u)) - mp.sqrt(u)) / mp.sqrt(u) ** 3) if u > 0 else -one / 6
    co = cheb_fit(g, mp.mpf(0), lim, 9)
    r = np.linspace(1e-9, np.pi / 2, 20001)
    got = r + r ** 3 * horner64(co, r * r)
    want = np.array([float(mp.sin(mp.mpf(float(x)))) for x in r])
    print(f"// sin |r|<=pi/2 deg 9: max rel err {np.max(np.abs(got - want) / want):.3e}")
    show("SIN64H: sin r = r + r^3 S(r^2), |r| <= pi/2, S degree 9", co)
    # kernels on |r| <= pi/4
    lim = (mp.pi / 4 + mp.mpf("0.001")) ** 2
    co = cheb_fit(g, mp.mpf(0), lim, 6)
    r = np.linspace(1e-9, np.pi / 4, 20001)
    got = r + r ** 3 * horner64(co, r * r)
    want = np.array([float(mp.sin(mp.mpf(float(x)))) for x in r])
    print(f"// sin |r|<=pi/4 deg 6: max rel err {np.max(np.abs(got - want) / want):.3e}")
    show("SIN64Q: sin r = r + r^3 S(r^2), |r| <= pi/4, S degree 6", co)
    hc = lambda u: ((mp.cos(mp.sqrt(u)) - 1 + u / 2) / u ** 2) if u > 0 else one / 24
    co = cheb_fit(hc, mp.mpf(0), lim, 6)
    got = 1 - r * r / 2 + r ** 4 * horner64(co, r * r)
    want = np.array([float(mp.cos(mp.mpf(float(x)))) for x in r])
    print(f"// cos |r|<=pi/4 deg 6: max rel err {np.max(np.abs(got - want) / want):.3e}")
    show("COS64Q: cos r = 1 - r^2/2 + r^4 C(r^2), |r| <= pi/4, C degree 6", co)
    for name, val in (("pi", mp.pi), ("pi/2", mp.pi / 2)):
        hi = mp.mpf(float(val))
        mid = mp.mpf(float(val - hi))
        lo = mp.mpf(float(val - hi - mid))
        print(f"// {name}: hi {mp.nstr(hi, 25)}  mid {mp.nstr(mid, 25)}  lo {mp.nstr(lo, 25)}")
    print(f"// 1/pi {mp.nstr(1 / mp.pi, 25)}   2/pi {mp.nstr(2 / mp.pi, 25)}")
