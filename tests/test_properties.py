"""Size-independent physical properties of the 7-DoF planar model, used as parity checks that
need no reference values: the body-frame dynamics do not depend on where the vehicle is or which
way it points (vehicle_model.py:376-382 read x, y, yaw nowhere), and the model is mirror-symmetric
about its longitudinal axis when left and right parameters agree.  The CPU half runs on the
oracle (hypothesis-driven); the GPU half runs the same properties on the HIP kernels."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

RW = 0.308309813617345
MIRROR_SIGN = np.array([1, -1, -1, 1, 1, 1, 1, -1, 1, -1, 1, -1.0])   # U V wz wFL wFR wRL wRR yaw x y ax ay
MIRROR_PERM = [0, 1, 2, 4, 3, 6, 5, 7, 8, 9, 10, 11]                   # swap left <-> right wheels


def mirror(s12):
    return s12[MIRROR_PERM] * MIRROR_SIGN[:, None]


def make_states(rng, n):
    s = np.zeros((12, n))
    s[0] = rng.uniform(8, 32, n)
    s[1] = rng.normal(0, 0.3, n)
    s[2] = rng.normal(0, 0.15, n)
    s[3:7] = s[0] / RW * (1 + rng.uniform(-0.01, 0.01, (4, n)))
    s[7] = rng.uniform(-3, 3, n)
    s[8:10] = rng.uniform(-50, 50, (2, n))
    s[10:12] = rng.normal(0, 1.0, (2, n))
    return s


def make_ctrl(rng, H, n):
    c = np.empty((H, 2, n))
    c[:, 0] = rng.uniform(-0.2, 0.2, (1, n)) * np.ones((H, 1))
    c[:, 1] = rng.uniform(-200, 400, (1, n)) * np.ones((H, 1))
    return c


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1), dx=st.floats(-1e3, 1e3), dy=st.floats(-1e3, 1e3),
       dpsi=st.floats(-6.0, 6.0))
def test_oracle_pose_invariance(oracle, seed, dx, dy, dpsi):
    rng = np.random.default_rng(seed)
    s0, ctrl = make_states(rng, 8), make_ctrl(rng, 30, 8)
    p = oracle.default_params()
    a = oracle.rollout(p, s0, ctrl, 1e-3)
    s1 = s0.copy()
    s1[7] += dpsi
    s1[8] += dx
    s1[9] += dy
    b = oracle.rollout(p, s1, ctrl, 1e-3)
    body = [0, 1, 2, 3, 4, 5, 6, 10, 11]
    assert np.array_equal(a[body], b[body]), "body-frame rows must not see the pose at all"
    # the displacement is the same vector, rotated by dpsi
    da, db = a[8:10] - s0[8:10], b[8:10] - s1[8:10]
    c, s = np.cos(dpsi), np.sin(dpsi)
    rot = np.stack([c * da[0] - s * da[1], s * da[0] + c * da[1]])
    assert np.abs(rot - db).max() <= 1e-9 * max(1.0, abs(dx), abs(dy))
    assert np.abs((b[7] - s1[7]) - (a[7] - s0[7])).max() <= 1e-12


@settings(max_examples=25, deadline=None)
@given(seed=st.integers(0, 2 ** 31 - 1))
def test_oracle_mirror_symmetry(oracle, seed):
    rng = np.random.default_rng(seed)
    s0, ctrl = make_states(rng, 8), make_ctrl(rng, 30, 8)
    p = oracle.default_params()
    a = oracle.rollout(p, s0, ctrl, 1e-3)
    cm = ctrl.copy()
    cm[:, 0] *= -1.0
    b = oracle.rollout(p, mirror(s0), cm, 1e-3)
    want = mirror(a)
    scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-6)
    assert (np.abs(b - want) <= 1e-10 * scale).all()


@pytest.mark.gpu
def test_gpu_pose_invariance_and_mirror_symmetry(gpu_vm):
    rng = np.random.default_rng(77)
    n, H = 4096, 100
    s0, ctrl = make_states(rng, n), make_ctrl(rng, H, n)
    for dtype, tol in ((np.float64, 1e-10), (np.float32, 2e-4)):
        vm = gpu_vm(1e-3)
        a = vm.rollout(s0.astype(dtype), ctrl.astype(dtype))
        s1 = s0.copy()
        s1[7] += rng.uniform(-6, 6, n)
        s1[8:10] += rng.uniform(-500, 500, (2, n))
        b = vm.rollout(s1.astype(dtype), ctrl.astype(dtype))
        body = [0, 1, 2, 3, 4, 5, 6, 10, 11]
        assert np.array_equal(a[body], b[body])
        cm = ctrl.copy()
        cm[:, 0] *= -1.0
        m = vm.rollout(mirror(s0).astype(dtype), cm.astype(dtype))
        want = mirror(a.astype(np.float64))
        scale = np.maximum(np.abs(want).max(axis=1, keepdims=True), 1e-6)
        assert (np.abs(m - want) <= tol * scale).all()
