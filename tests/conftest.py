import importlib
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def pkg():
    return importlib.import_module("python-motionplanning_amd")


@pytest.fixture(scope="session")
def workloads(pkg):
    return pkg.workloads


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def rel_err(a, b, floor=1e-9):
    """max |a-b| / max(|b|, floor-scaled magnitude of the row)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def parity(got, want, tol, what=""):
    """The parity bar of BASELINE.json's north_star: relative error `tol`
    (1e-6 fp64, 1e-3 fp32) per state row, |got - want| <= tol * max(|want|, row scale),
    where the row scale is max|want| over the row (so rows that cross zero are
    judged against their own magnitude).  `got`, `want`: [rows][N] (or [..., rows, N])."""
    g = np.asarray(got, dtype=np.float64)
    w = np.asarray(want, dtype=np.float64)
    assert g.shape == w.shape, (g.shape, w.shape)
    assert np.isfinite(w).all(), "reference values must be finite"
    assert np.isfinite(g).all(), f"{what}: non-finite result"
    if g.size == 0:
        return 0.0
    scale = np.maximum(np.abs(w).max(axis=-1, keepdims=True), 1e-30)
    err = np.abs(g - w) / np.maximum(np.abs(w), scale)
    worst = float(err.max())
    assert worst <= tol, f"{what}: relative error {worst:.3e} > {tol:g}"
    return worst


def parity_elementwise(got, want, tol, floor_frac=1e-2, what=""):
    """The stricter reading of the same bar: every ELEMENT that is at least `floor_frac` of its row's scale
    is held to `tol` relative to ITSELF (|got - want| <= tol * max(|want|, floor_frac * row scale)); smaller
    elements -- rows that cross zero -- are held to the absolute floor tol * floor_frac * row scale.
    With floor_frac = 1e-2 that is 100 times tighter on small entries than `parity`."""
    g = np.asarray(got, dtype=np.float64)
    w = np.asarray(want, dtype=np.float64)
    assert g.shape == w.shape and np.isfinite(w).all() and np.isfinite(g).all(), what
    if g.size == 0:
        return 0.0
    scale = np.maximum(np.abs(w).max(axis=-1, keepdims=True), 1e-30)
    err = np.abs(g - w) / np.maximum(np.abs(w), floor_frac * scale)
    worst = float(err.max())
    assert worst <= tol, f"{what}: element-wise relative error {worst:.3e} > {tol:g} (floor {floor_frac:g} of the row scale)"
    return worst


@pytest.fixture(scope="session")
def gpu_vm(pkg):
    """VehicleModel factory on cuda:0; skips nothing: -m gpu tests REQUIRE the HIP path."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"

    def make(dt, **kw):
        return pkg.VehicleModel(2.906, np.deg2rad(30), dt, device=0, **kw)
    return make
