"""CPU: the Python shim's packing / unpacking of the single-vehicle drop-ins and of the batched
rollout call, exercised end to end against a TEST-ONLY stand-in for the C ABI handle that
answers with the oracle (the product has no CPU path; this checks the host logic only:
persistent I/O buffers, pointer offsets, return-list layout, parameter caching)."""
import ctypes as C
import importlib

import numpy as np
import pytest


def _view(ptr, n, dtype=np.float64):
    addr = ptr.value if isinstance(ptr, C.c_void_p) else ptr
    return np.ctypeslib.as_array((C.c_double * n).from_address(addr)) if dtype == np.float64 else None


class OracleBackedHandle:
    def __init__(self, oracle):
        self.O = oracle
        self.p = oracle.default_params()
        self.calls = []
        self.param_sets = 0
        import threading
        self.lock = threading.RLock()           # what _lib.Handle carries (set_option / launch / reset as one step)
        self.rows_seen = []                     # (rows the option said at launch, rows of the caller's state)

    def set_params(self, cp, key=None):
        self.param_sets += 1
        for name in ("m", "a", "b", "Izz", "Jw", "hg", "T", "wL", "wR", "rw", "g"):
            setattr(self.p, name, getattr(cp, name))
        for i in range(4):
            self.p.B[i], self.p.C[i] = cp.B[i], cp.C[i]

    def call(self, name, *a):
        self.calls.append(name)
        if name == "vdyn_step_f64_host":
            n, s_in, ctrl, k, dt, mu4, s_out, sd, out = a
            assert n == 1 and k == 12 and mu4 is None
            s, c = _view(s_in, 12), _view(ctrl, 12)
            o = self.O.planar_model_RK4(self.p, dt, s[:10], c[4:8], c[8:12], c[0:4], s[10], s[11])
            _view(s_out, 12)[:] = np.concatenate([o[0], [o[7], o[8]]])
            _view(sd, 10)[:] = o[5]
            _view(out, 18)[:] = o[6]
        elif name == "vdyn_planar_model_f64_host":
            n, st, c12, accp, sd, aux, out, acc = a
            s, c, ap = _view(st, 10), _view(c12, 12), _view(accp, 2)
            o = self.O.planar_model(self.p, s, c[4:8], c[8:12], c[0:4], ap[0], ap[1])
            _view(sd, 10)[:] = o[0]
            _view(aux, 4)[:] = o[1:5]
            _view(out, 18)[:] = o[5]
            _view(acc, 2)[:] = o[6:8]
        elif name == "vdyn_rollout_spiral_f64_host":
            n, H, s0, sp, wheelbase, max_steer, torque, dt, mu4, term, traj, stride = a
            st = _view(s0, 12 * n).reshape(12, n)
            spv = _view(sp, 3 * n).reshape(n, 3)
            self.spiral_args = (n, H, wheelbase, max_steer, torque, dt, stride)
            out = self.O.rollout_spiral(self.p, st, spv, H, dt, wheelbase=wheelbase, max_steer=max_steer, torque=torque,
                                        traj_stride=stride if traj is not None and traj.value else 0)
            if isinstance(out, tuple):
                _view(term, 12 * n)[:] = out[0].ravel()
                _view(traj, out[1].size)[:] = out[1].ravel()
            else:
                _view(term, 12 * n)[:] = out.ravel()
        elif name == "vdyn_set_option":
            self.options = getattr(self, "options", []) + [tuple(a)]
            import time
            time.sleep(0)                       # invite a thread switch between "set" and "launch"
        elif name == "vdyn_rollout_f32_host":
            # records the marshalling only: rows of the state arrays follow the option set just before
            n, H, s0, ct, k, layout, pid, P, dt, mu4, term, traj, stride = a
            self.rollout_rows = [v for o, v in self.options if o == 2][-1] if getattr(self, "options", None) else 12
            rows = self.rollout_rows
            self.rows_seen.append((rows, getattr(self, "caller_rows", {}).get(n)))
            src = np.ctypeslib.as_array((C.c_float * (rows * n)).from_address(s0.value)).reshape(rows, n)
            np.ctypeslib.as_array((C.c_float * (rows * n)).from_address(term.value)).reshape(rows, n)[:] = src + 1.0
        else:
            raise AssertionError(f"unexpected ABI call {name}")


@pytest.fixture
def vm_mock(pkg, oracle):
    vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-4)
    h = OracleBackedHandle(oracle)
    vm._handles[0] = h
    return vm, h


def test_planar_model_RK4_return_list_and_chaining(vm_mock, pkg, oracle):
    from conftest import load_golden
    vm, h = vm_mock
    g = load_golden("g4_closed_loop_world.npz")
    p = pkg.VehicleParameters()
    state, ax, ay = list(g["state"][0]), 0, 0                      # a list first (drive.py:64), arrays after
    for i in range(40):
        o = vm.planar_model_RK4(state, list(g["torque"][i]), [1.0] * 4, list(g["delta"][i]), p, ax, ay)
        assert len(o) == 9 and o[0].shape == (10,) and o[5].shape == (10,) and o[6].shape == (18,)
        assert (o[1], o[2], o[3], o[4]) == (o[0][8], o[0][9], o[0][7], o[0][0])
        assert np.abs(o[0] - g["state_update"][i]).max() <= 1e-11 * np.abs(g["state_update"][i]).max()
        assert abs(o[7] - g["acc"][i][0]) <= 1e-10 and abs(o[8] - g["acc"][i][1]) <= 1e-10
        keep = o[0].copy()
        state, ax, ay = o[0], o[7], o[8]
        nxt = vm.planar_model_RK4(state, list(g["torque"][i]), [1.0] * 4, list(g["delta"][i]), p, ax, ay)
        assert np.array_equal(o[0], keep), "results of an earlier call must not be overwritten by the next"
        assert nxt[0] is not o[0]
    assert p.DFL == 1.0 and h.param_sets >= 1


def test_planar_model_return_list(vm_mock, pkg):
    from conftest import load_golden
    vm, _ = vm_mock
    g = load_golden("g2_deriv.npz")
    for i in (0, 9, 33):
        o = vm.planar_model(g["state"][i], g["torque"][i], g["mu"][i], g["delta"][i], pkg.VehicleParameters(),
                            *g["ax_ay_prev"][i])
        assert len(o) == 8 and o[0].shape == (10,) and o[5].shape == (18,)
        assert np.abs(o[0] - g["state_dot"][i]).max() <= 1e-11 * np.abs(g["state_dot"][i]).max()
        assert np.abs(np.array(o[1:5]) - g["aux"][i]).max() <= 1e-11 * np.abs(g["aux"][i]).max()
        assert np.abs(np.array(o[6:8]) - g["acc"][i]).max() <= 1e-11 * np.abs(g["acc"][i]).max()


def test_parameter_changes_are_noticed(vm_mock, pkg):
    vm, h = vm_mock
    p = pkg.VehicleParameters()
    st = [25.0, 0, 0] + [25.0 / p.rw] * 4 + [0, 0, 0]
    a = vm.planar_model_RK4(st, [100.0] * 4, [1.0] * 4, [0.05, 0.05, 0, 0], p, 0, 0)[0]
    n0 = h.param_sets
    vm.planar_model_RK4(st, [100.0] * 4, [1.0] * 4, [0.05, 0.05, 0, 0], p, 0, 0)
    p.BFL = p.BFR = 12.0                                           # softer front tires (vehicle_model.py:237-242)
    b = vm.planar_model_RK4(st, [100.0] * 4, [1.0] * 4, [0.05, 0.05, 0, 0], p, 0, 0)[0]
    assert h.param_sets > n0 and not np.array_equal(a, b)
    q = pkg.VehicleParameters(mf=1200.0)                           # a different object altogether
    c = vm.planar_model_RK4(st, [100.0] * 4, [1.0] * 4, [0.05, 0.05, 0, 0], q, 0, 0)[0]
    assert not np.array_equal(b, c)


def test_rollout_spiral_host_marshalling(vm_mock, oracle, workloads):
    """VehicleModel.rollout_spiral: [E][P][3] parameters are flattened ego-major, wheelbase / max_steer default to
    the model's (VehicleModel(wheelbase, max_steer, dt), drive.py:109), shapes are validated before any call."""
    vm, h = vm_mock
    s0, sp = workloads.config3_spiral(21, 10, np.float64)
    term = vm.rollout_spiral(s0, sp.reshape(3, 7, 3), 30, torque=80.0, dt=1e-3)
    n, H, wheelbase, max_steer, torque, dt, stride = h.spiral_args
    assert (n, H, torque, dt, stride) == (21, 30, 80.0, 1e-3, 0)
    assert wheelbase == 2.906 and abs(max_steer - np.deg2rad(30)) < 1e-15
    want = oracle.rollout_spiral(oracle.default_params(), s0, sp, 30, 1e-3, torque=80.0)
    assert np.array_equal(term, want)
    term2, traj = vm.rollout_spiral(s0, sp, 30, dt=1e-3, traj_stride=10, max_steer=0.05)
    assert traj.shape == (3, 12, 21) and np.array_equal(traj[-1], term2) and h.spiral_args[3] == 0.05
    for bad in (lambda: vm.rollout_spiral(s0, sp[:-1], 5), lambda: vm.rollout_spiral(s0[:10], sp, 5),
                lambda: vm.rollout_spiral(s0, sp, -1), lambda: vm.rollout_spiral(s0, sp.reshape(7, 3, 3)[:, :, :2], 5)):
        with pytest.raises(ValueError):
            bad()


def test_rollout_state_rows_option_marshalling(vm_mock, pkg, workloads):
    """VehicleModel.rollout with a [22][N] fp32 state (include/vdyn.h, VDYN_OPT_STATE_ROWS): the option is set to 22
    for that call and back to 12 after it -- also when the call raises --, the terminal state has 22 rows; fp64
    and other row counts are refused before any ABI call."""
    vm, h = vm_mock
    lib = pkg._lib
    s0, tab, pid = workloads.config3(70, 6, np.float32)
    s22 = np.concatenate([s0, np.zeros((10, 70), np.float32)])
    ctrl = workloads.expand_shared_controls(tab, pid)
    term = vm.rollout(s22, ctrl)
    assert term.shape == (22, 70) and term.dtype == np.float32 and np.array_equal(term, s22 + 1.0)
    assert h.options == [(lib.VDYN_OPT_STATE_ROWS, 22), (lib.VDYN_OPT_STATE_ROWS, 12)] and h.rollout_rows == 22
    h.options = []
    term = vm.rollout(s0, ctrl)
    assert term.shape == (12, 70) and h.options == [] and h.rollout_rows == 12
    n_calls = len(h.calls)
    for bad in (lambda: vm.rollout(s22.astype(np.float64), ctrl.astype(np.float64)),
                lambda: vm.rollout(s22[:17], ctrl), lambda: vm.rollout(s22[:11], ctrl)):
        with pytest.raises(ValueError):
            bad()
    assert len(h.calls) == n_calls, "refused before any ABI call"

    def boom(name, *a):
        if name.startswith("vdyn_rollout"):
            raise pkg.VdynError(-1, "simulated")
        return OracleBackedHandle.call(h, name, *a)
    h.options = []
    h.call, orig = boom, h.call
    with pytest.raises(pkg.VdynError):
        vm.rollout(s22, ctrl)
    h.call = orig
    assert h.options[-1] == (lib.VDYN_OPT_STATE_ROWS, 12), "the option is restored when the call fails"


def test_interpolate_waypoints_out_argument_is_validated(vm_mock):
    vm, _ = vm_mock
    paths = np.zeros((2, 7, 3, 49))
    best = np.zeros(2, np.int32)
    for wp, wc in ((np.zeros((2, 100, 2), np.float32), np.zeros(2, np.int32)),      # dtype of the call is float64
                   (np.zeros((2, 99, 2)), np.zeros(2, np.int32)),                    # Wmax mismatch
                   (np.zeros((2, 100, 2)), np.zeros(2, np.int64)),                   # counts must be int32
                   (np.zeros((2, 100, 2)), np.zeros(3, np.int32))):
        with pytest.raises(ValueError):
            vm.interpolate_waypoints(paths, best, 0.01, 100, out=(wp, wc))


def test_dropin_warns_like_numpy_on_nonfinite_results(vm_mock, pkg):
    """The reference propagates inf / nan and NumPy raises a RuntimeWarning (vehicle_model.py:284-293: division
    by a zero wheel-plane speed); so does the drop-in."""
    import warnings
    vm, _ = vm_mock
    p = pkg.VehicleParameters()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        o = vm.planar_model_RK4([0.0] * 10, [0.0] * 4, [1.0] * 4, [0.0] * 4, p, 0.0, 0.0)
    assert not np.isfinite(o[0]).all() and any(issubclass(x.category, RuntimeWarning) for x in w)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        vm.planar_model_RK4([25.0, 0, 0] + [25.0 / p.rw] * 4 + [0, 0, 0], [0.0] * 4, [1.0] * 4, [0.0] * 4, p, 0.0, 0.0)
    assert not [x for x in w if "planar_model_RK4" in str(x.message)]


def test_state_rows_option_is_not_shared_between_threads(vm_mock, workloads):
    """VDYN_OPT_STATE_ROWS is handle state; two threads on one VehicleModel -- one with [12][N], one
    with [22][N] states -- must each launch with their own row count (a 12-row call launched while the option says 22
    reads and writes 22 rows of 12-row buffers)."""
    import threading
    vm, h = vm_mock
    n12, n22 = 14, 21                                       # the batch size tells the mock whose call it is
    h.caller_rows = {n12: 12, n22: 22}
    _, tab, _ = workloads.config3(7, 5, np.float32)
    pid12, pid22 = (np.arange(n12) % 7).astype(np.int32), (np.arange(n22) % 7).astype(np.int32)
    s12 = np.zeros((12, n12), np.float32)
    s22 = np.zeros((22, n22), np.float32)
    errs = []

    def work(s, pid):
        try:
            for _ in range(150):
                out = vm.rollout(s, tab, path_id=pid)
                assert out.shape == s.shape
        except Exception as e:                              # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=work, args=a) for a in ((s12, pid12), (s22, pid22))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs
    assert len(h.rows_seen) == 300 and all(opt == mine for opt, mine in h.rows_seen), \
        [x for x in h.rows_seen if x[0] != x[1]][:5]
