"""GPU (-m gpu): the bounded-range elementary functions of the FAST step (csrc/vdyn_fastmath.hpp scalar
forms, csrc/vdyn_packed.hpp packed forms) evaluated ON THE DEVICE through vdyn_fastmath_eval_* on dense
grids, against float64 libm.  These are the accuracy statements the kernels' comments make."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

U32, U64 = 2.0 ** -24, 2.0 ** -53          # half an ulp of 1.0: "n ulp" below = 2 n U relative


def _grid(lo, hi, n, log=False, both=False):
    g = np.geomspace(lo, hi, n) if log else np.linspace(lo, hi, n)
    return np.concatenate([-g[::-1], g]) if both else g


def _rel(got, want, floor=0.0):
    return np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want), floor)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_scalar_forms(gpu_vm, dtype):
    vm = gpu_vm(1e-3)
    u = U32 if dtype == np.float32 else U64
    ev = lambda fn, x, c=0.0: [np.asarray(o, np.float64) for o in vm.fastmath_eval(fn, x.astype(dtype), c)]
    x = _grid(1e-6, 1e6, 200000, log=True, both=True).astype(dtype)
    got = ev(0, x)[0]
    e = _rel(got, np.arctan(x.astype(np.float64))).max()
    assert e <= 4.5 * u, f"atan_rcp: {e / u:.2f} U"                               # 2 ulp (+ 1/x by v_rcp: 1 ulp)
    x = _grid(0.0, np.pi, 400001).astype(dtype)
    s = np.sin(x.astype(np.float64))
    got = ev(1, x)[0]
    assert np.abs(got - s).max() <= 4 * u and _rel(got, s, 1e-300)[x < 3.0].max() <= 4.5 * u     # sin_0_pi
    lim = 1e4 if dtype == np.float32 else 1e9
    x = np.concatenate([_grid(-40.0, 40.0, 200001), _grid(1.0, lim, 100000, log=True, both=True)]).astype(dtype)
    got = ev(2, x)[0]
    assert np.abs(got - np.sin(x.astype(np.float64))).max() <= 6 * u, "sin_mid"
    lim = 65536.0 if dtype == np.float32 else 2.0 ** 30
    x = np.concatenate([_grid(-50.0, 50.0, 200001), _grid(1.0, lim, 100000, log=True, both=True)]).astype(dtype)
    gs, gc = ev(3, x)
    es, ec = np.abs(gs - np.sin(x.astype(np.float64))).max(), np.abs(gc - np.cos(x.astype(np.float64))).max()
    assert es <= 5 * u and ec <= 5 * u, f"sincos_mid {es / u:.2f} {ec / u:.2f} U"
    x = _grid(-np.pi / 4, np.pi / 4, 400001).astype(dtype)
    gs, gc = ev(4, x)
    assert _rel(gs, np.sin(x.astype(np.float64)), 1e-300).max() <= 3 * u and np.abs(gc - np.cos(x.astype(np.float64))).max() <= 3 * u
    print(f"\n  {np.dtype(dtype).name}: atan {e / u:.2f} U, sincos_mid {es / u:.2f} / {ec / u:.2f} U")


def test_packed_forms_of_the_fp32_step(gpu_vm):
    vm = gpu_vm(1e-3)
    ev = lambda fn, x, c=0.0: [np.asarray(o, np.float64) for o in vm.fastmath_eval(fn, x.astype(np.float32), c)]
    # the tire chain: sin(C atan x) = x c W_C(c), c = rsq(1 + x^2), with the handle's own fit of W_C -- every x, either sign
    x = np.concatenate([[0.0], _grid(-4.0, 4.0, 200001), _grid(1e-8, 1e7, 100000, log=True, both=True)]).astype(np.float32)
    xd = x.astype(np.float64)
    worst = worst_g = 0.0
    for c in (1.5047, 0.25, 1.0, 1.3, 1.9, 2.0):
        mu, g = ev(5, x, c)
        worst = max(worst, np.abs(mu - np.sin(c * np.arctan(xd))).max())
        small = np.abs(xd) <= np.sqrt(3.0)
        gref = np.where(xd != 0, np.sin(c * np.arctan(xd)) / np.where(xd != 0, xd, 1.0), c)
        worst_g = max(worst_g, (np.abs(g - gref) / np.abs(gref))[small].max())
        assert abs(g[0] - c) <= 5e-7 * c                   # G(0) = C: quirk Q5 (s == 0) needs no branch
    assert worst <= 5e-7, f"sin(C atan x): {worst:.2e}"
    assert worst_g <= 5e-7, f"sin(C atan x) / x, relative for x <= sqrt(3): {worst_g:.2e}"
    with pytest.raises(Exception):
        vm.fastmath_eval(5, x, 3.0)                        # no validated fit for this shape factor
    # unwrapped yaw: |x| <= 2^16
    x = np.concatenate([_grid(-50.0, 50.0, 200001), _grid(1.0, 65536.0, 100000, log=True, both=True)]).astype(np.float32)
    gs, gc = ev(6, x)
    ey = max(np.abs(gs - np.sin(x.astype(np.float64))).max(), np.abs(gc - np.cos(x.astype(np.float64))).max())
    assert ey <= 4e-7, f"sincos of yaw: {ey:.2e}"
    # stage rotation: |d| <= 1/32
    x = _grid(-1.0 / 32, 1.0 / 32, 200001).astype(np.float32)
    gs, gc = ev(7, x)
    assert _rel(gs, np.sin(x.astype(np.float64)), 1e-300).max() <= 2.0e-7 and np.abs(gc - np.cos(x.astype(np.float64))).max() <= 1.2e-7
    # steering kernel: |delta| <= pi/4
    x = _grid(-np.pi / 4, np.pi / 4, 200001).astype(np.float32)
    gs, gc = ev(8, x)
    assert _rel(gs, np.sin(x.astype(np.float64)), 1e-300).max() <= 2.5e-7 and np.abs(gc - np.cos(x.astype(np.float64))).max() <= 2.0e-7
    print(f"\n  packed: sin(C atan x) abs {worst:.2e}, G relative {worst_g:.2e}, yaw sincos abs {ey:.2e}")
    with pytest.raises(Exception):
        vm.fastmath_eval(6, x.astype(np.float64))          # packed forms exist in fp32 only


def test_fitted_tire_chain_fp64(gpu_vm):
    """fn 5 in fp64: the trimmed scalar step's chain, c = rsq(1 + x^2), degree-16 Horner with the fit of C (2.3e-14;
    degree 18 held 1.3e-15 for two more fmas per wheel and stage: profiles/r04_tire_fit_degrees.txt)."""
    vm = gpu_vm(1e-3)
    x = np.concatenate([[0.0], _grid(-4.0, 4.0, 200001), _grid(1e-8, 1e7, 100000, log=True, both=True)])
    xl = x.astype(np.longdouble)
    worst = worst_g = 0.0
    for c in (1.5047, 0.25, 1.0, 1.3, 1.9, 2.0):
        mu, g = [np.asarray(o, np.float64).astype(np.longdouble) for o in vm.fastmath_eval(5, x, c)]
        worst = max(worst, float(np.abs(mu - np.sin(c * np.arctan(xl))).max()))
        small = np.abs(xl) <= np.sqrt(3.0)
        gref = np.where(xl != 0, np.sin(c * np.arctan(xl)) / np.where(xl != 0, xl, 1.0), c)
        worst_g = max(worst_g, float((np.abs(g - gref) / np.abs(gref))[small].max()))
    assert worst <= 5e-14 and worst_g <= 5e-14, f"{worst:.2e} {worst_g:.2e}"
    print(f"\n  fp64 fitted chain: sin(C atan x) abs {worst:.2e}, G relative {worst_g:.2e}")


def test_nonfinite_lanes_status(gpu_vm, workloads):
    """vdyn_nonfinite_lanes_*: the batched counterpart of NumPy's RuntimeWarning in the reference."""
    import warnings
    s0, ctrl = workloads.config2(8, 20)
    s0[0, 5] = 0.0                                        # U = 0 and wheels at rest: vx = 0 -> division by zero
    s0[3:7, 5] = 0.0
    s0[0, 11] = np.nan
    vm = gpu_vm(1e-3)
    term = vm.rollout(s0, ctrl)
    st, cnt = vm.nonfinite_lanes(term)
    assert cnt == 2 and st.tolist() == [1 if i in (5, 11) else 0 for i in range(64)]
    import torch
    st_d, cnt_d = vm.nonfinite_lanes(torch.from_numpy(term.astype(np.float32)).to("cuda:0"))
    assert cnt_d == 2 and np.array_equal(st_d.cpu().numpy(), st)
    assert vm.nonfinite_lanes(np.zeros((12, 0)))[1] == 0
    # the single-vehicle drop-in warns like the reference does
    vm1 = gpu_vm(1e-4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        vm1.planar_model_RK4([0.0] * 10, [0.0] * 4, [1.0] * 4, [0.0] * 4, vm1.params, 0.0, 0.0)
    assert any(issubclass(x.category, RuntimeWarning) for x in w)
