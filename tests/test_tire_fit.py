"""The per-handle tire fit of the fp32 step (include/vdyn.h, vdyn_tire_fit_f32): host arithmetic, no GPU.

sin(C atan x) / x = c W_C(c) with c = 1 / sqrt(1 + x^2): the coefficients the library would hand its kernels are
evaluated here the way the kernel evaluates them (fp32 Horner in c, one rounding per fma) and compared with
float64 NumPy over every slip, for the shape factors a handle may carry.  The device evaluation of the same
chain is checked in tests/test_gpu_fastmath.py (fn 5)."""
import numpy as np
import pytest


def _horner32(coef, c):
    # float64 product + sum of two floats rounded once to float32 is the fma's result
    g = np.full_like(c, coef[0])
    for a in coef[1:]:
        g = (g.astype(np.float64) * c.astype(np.float64) + np.float64(a)).astype(np.float32)
    return g


@pytest.mark.parametrize("C", [1.5047, 0.0, 0.1, 0.5, 1.0, 1.3, 1.7, 1.9, 2.0])
def test_fit_matches_pacejka_shape_function_for_every_slip(pkg, C):
    coef, ok = pkg.VehicleModel.tire_fit(C)
    assert ok and coef.dtype == np.float32 and coef.shape == (9,)
    x = np.concatenate([[0.0], np.linspace(0.0, 4.0, 200001), np.geomspace(1e-8, 1e8, 100001)])
    c = (1.0 / np.sqrt((1.0 + (x * x).astype(np.float32)).astype(np.float32).astype(np.float64))).astype(np.float32)
    G = (_horner32(coef, c).astype(np.float64) * c.astype(np.float64)).astype(np.float32).astype(np.float64)
    want = np.where(x > 0, np.sin(C * np.arctan(x)) / np.where(x > 0, x, 1.0), C)
    assert np.max(np.abs(G - want) * x) <= 5e-7                       # mu / D = sin(C atan x), absolute
    small = x <= np.sqrt(3.0)
    assert np.max((np.abs(G - want) / np.maximum(np.abs(want), 1e-30))[small]) <= 5e-7   # stiffness at small slip, relative
    assert G[-1] <= 1e-8 and abs(G[0] - C) <= 5e-7 * max(C, 1e-30)   # x -> inf: 0; x = 0: C (quirk Q5 without a branch)


def test_integer_shape_factors_are_chebyshev_polynomials(pkg):
    # W_1 = 1, W_2 = 2 c (U_0, U_1): the fit must reproduce them to rounding
    c1, ok1 = pkg.VehicleModel.tire_fit(1.0)
    c2, ok2 = pkg.VehicleModel.tire_fit(2.0)
    assert ok1 and ok2
    assert np.allclose(c1, [0] * 8 + [1], atol=2e-7) and np.allclose(c2, [0] * 7 + [2, 0], atol=4e-7)


@pytest.mark.parametrize("C", [3.0, 4.5, float("nan"), float("inf")])
def test_fit_that_fails_its_check_is_reported(pkg, C):
    _, ok = pkg.VehicleModel.tire_fit(C)
    assert not ok          # such a handle keeps the atan -> sine chain (lane_cs in csrc/vdyn_kernels.hip)


@pytest.mark.parametrize("C", [1.5047, 0.0, 0.5, 1.0, 1.3, 1.9, 2.0])
def test_fp64_fit(pkg, C):
    """Degree 16, 17 coefficients (round 3: degree 18; two fmas per wheel and stage bought 1.3e-15 instead of 2.3e-14,
    profiles/r04_tire_fit_degrees.txt): checked to 5e-14 by the library, here against long-double NumPy with a plain
    (unfused) Horner loop."""
    coef, ok = pkg.VehicleModel.tire_fit(C, np.float64)
    assert ok and coef.dtype == np.float64 and coef.shape == (17,)
    x = np.concatenate([[0.0], np.linspace(0.0, 4.0, 100001), np.geomspace(1e-8, 1e8, 50001)]).astype(np.longdouble)
    c = (1.0 / np.sqrt(1.0 + x * x)).astype(np.float64)
    g = np.full_like(c, coef[0])
    for a in coef[1:]:
        g = g * c + a
    G = (g * c).astype(np.longdouble)
    want = np.where(x > 0, np.sin(C * np.arctan(x)) / np.where(x > 0, x, 1.0), C)
    assert np.max(np.abs(G - want) * x) <= 5e-14
    small = x <= np.sqrt(3.0)
    assert np.max((np.abs(G - want) / np.maximum(np.abs(want), 1e-300))[small]) <= 5e-14
