"""Host-side mirror of the reference's vehicle-model interface, executed on MI355X.

``VehicleModel`` keeps the constructor, method names, argument meaning and
return lists of /root/reference/libs/vehicle_model/vehicle_model.py
(``VehicleModel.__init__`` :69-95, ``planar_model`` :220-425,
``planar_model_RK4`` :427-445) so that drive.py:109 / drive.py:141-143 can use
it unchanged, and adds the batched entry points BASELINE.json's north_star
names: ``step``, ``rollout`` and ``mpc_argmin``.

Every method runs the hand-written HIP kernels behind ``include/vdyn.h``.
There is no NumPy or CPU implementation of the model in this package.

Batched arrays are struct-of-arrays, ``[rows][N]`` (see include/vdyn.h):
``state12`` rows ``U,V,wz,wFL,wFR,wRL,wRR,yaw,x,y,ax_prev,ay_prev``.
NumPy inputs go through the ``_host`` ABI (staged copies, synchronous);
torch CUDA tensors go through the ``_dev`` ABI zero-copy on torch's current
stream (asynchronous, like any torch op).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import VdynError, VdynParams  # noqa: F401  (re-exported)

_WHEELS = ("FL", "FR", "RL", "RR")


class VehicleParameters:
    """Same constructor arguments and attribute names as the reference's
    ``VehicleParameters`` (vehicle_model.py:17-61), so objects of either class
    can be passed as ``p``."""

    def __init__(self, mf=987.89, mr=869.93, mus=50, L=2.906, ab_ratio=0.85, T=1.536,
                 hg=0.55419, Jw=1, kf=26290, kr=25830, Efront=0.0376, Erear=0,
                 LeverArm=0.13256, BFL=20.6357, CFL=1.5047, DFL=1.1233):
        self.rr = 0.329
        self.mus, self.mf, self.mr = mus, mf, mr
        self.m = mf + mr
        self.L, self.ab_ratio = L, ab_ratio
        self.b = L / (1 + ab_ratio)
        self.a = L - self.b
        self.Izz = 0.5 * self.m * self.a * self.b
        self.Jw, self.hg, self.T = Jw, hg, T
        self.kf, self.kr = kf, kr
        self.rw = self.rr - (mf / 2 + mus) / kf
        for w in _WHEELS:  # one tire model on all four corners (:41-54)
            setattr(self, "B" + w, BFL)
            setattr(self, "C" + w, CFL)
            setattr(self, "D" + w, DFL)
        self.Efront, self.Erear = Efront, Erear
        self.E = [Efront, Efront, Erear, Erear]
        self.LeverArm = LeverArm
        self.wL = self.wR = T / 2


def params_to_c(p) -> VdynParams:
    """Any object with the reference's VehicleParameters attributes -> VdynParams."""
    c = VdynParams()
    for n in ("m", "a", "b", "Izz", "Jw", "hg", "T", "wL", "wR", "rw"):
        setattr(c, n, float(getattr(p, n)))
    c.g = 9.81  # vehicle_model.py:230
    for i, w in enumerate(_WHEELS):
        c.B[i] = float(getattr(p, "B" + w))
        c.C[i] = float(getattr(p, "C" + w))
    return c


def _is_torch_cuda(x):
    return hasattr(x, "data_ptr") and getattr(x, "is_cuda", False)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float64:
        return "f64"
    if dtype == np.float32:
        return "f32"
    raise ValueError(f"unsupported dtype {dtype}: float32 or float64 only")


def _vp(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())


class _Backend:
    """Uniform view over NumPy (host ABI) and torch-CUDA (device ABI) buffers."""

    def __init__(self, like, dtype=None):
        self.torch = _is_torch_cuda(like)
        if self.torch:
            import torch
            self._t = torch
            self.device = like.device
            self.np_dtype = np.dtype({torch.float32: np.float32, torch.float64: np.float64}[like.dtype]) \
                if like.dtype in (torch.float32, torch.float64) else None
            if self.np_dtype is None:
                raise ValueError("unsupported tensor dtype: float32 or float64 only")
            self.t_dtype = like.dtype
        else:
            self.np_dtype = np.dtype(dtype or np.asarray(like).dtype)
            if self.np_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
                self.np_dtype = np.dtype(np.float64)
        self.suffix = _suffix(self.np_dtype)
        self.kind = "dev" if self.torch else "host"

    def inp(self, x, shape=None, int32=False):
        """Contiguous input buffer of the call's dtype (never modified)."""
        if x is None:
            return None
        if self.torch:
            want = self._t.int32 if int32 else self.t_dtype
            if not _is_torch_cuda(x):
                x = self._t.as_tensor(np.asarray(x), device=self.device)
            if x.device != self.device:
                raise ValueError("all tensors of one call must live on the same device")
            x = x.to(want).contiguous()
        else:
            x = np.ascontiguousarray(x, dtype=np.int32 if int32 else self.np_dtype)
        if shape is not None and tuple(x.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(x.shape)}")
        return x

    def out(self, *shape, int32=False, reuse=None):
        """A fresh output buffer, or `reuse` -- the caller's own (same kind, dtype, shape, C-contiguous): a large host
        array that lives across calls spares every call the page faults of a fresh np.empty (1.2 GB of DataLog is
        290 000 pages)."""
        if reuse is not None:
            if self.torch:
                ok = _is_torch_cuda(reuse) and reuse.device == self.device and reuse.is_contiguous() and \
                    reuse.dtype == (self._t.int32 if int32 else self.t_dtype)
            else:
                ok = isinstance(reuse, np.ndarray) and reuse.flags.c_contiguous and reuse.flags.writeable and \
                    reuse.dtype == (np.int32 if int32 else self.np_dtype)
            if not ok or tuple(reuse.shape) != tuple(shape):
                raise ValueError(f"out: need a writable C-contiguous {'tensor on the inputs device' if self.torch else 'ndarray'} "
                                 f"of shape {tuple(shape)} and the call's dtype")
            return reuse
        if self.torch:
            return self._t.empty(shape, dtype=self._t.int32 if int32 else self.t_dtype, device=self.device)
        return np.empty(shape, dtype=np.int32 if int32 else self.np_dtype)

    def stream_args(self):
        if not self.torch:
            return ()
        return (C.c_void_p(self._t.cuda.current_stream(self.device).cuda_stream),)

    def device_index(self, default):
        if self.torch:
            return self.device.index if self.device.index is not None else self._t.cuda.current_device()
        return default


class VehicleModel:
    """Drop-in for the reference ``VehicleModel`` on the RK4 / Pacejka path.

    ``VehicleModel(wheelbase, max_steer, dt)`` as at drive.py:109; only ``dt``
    matters to this path (vehicle_model.py:93,:428).  ``device`` selects the
    GPU used for NumPy inputs (torch tensors bring their own device).  ``lanes_per_rollout``:
    1 (default) one lane per rollout; 4 wheel-parallel (for N <= 16384, agrees to rounding);
    0 automatic (include/vdyn.h, VDYN_OPT_LANES_PER_ROLLOUT)."""

    def __init__(self, wheelbase=1.0, max_steer=0.7, dt=0.05, device=0, params=None, lanes_per_rollout=1):
        if lanes_per_rollout not in (0, 1, 4):
            raise ValueError("lanes_per_rollout: 1 (lane per rollout), 4 (wheel-parallel) or 0 (automatic)")
        self.lanes_per_rollout = int(lanes_per_rollout)
        self.dt = dt
        self.wheelbase = wheelbase
        self.max_steer = max_steer
        self.device = int(device)
        self.params = params if params is not None else VehicleParameters()
        self._handles = {}
        _lib.load()  # fail now, loudly, if the HIP library is not built

    def handle(self, device=None):
        """The library handle of `device` (default: this model's device), e.g. for
        ``distributed.PeerExchange``."""
        return self._handle(self.device if device is None else int(device))

    # ------------------------------------------------------------------ plumbing
    def _handle(self, device, p=None):
        cp = params_to_c(p if p is not None else self.params)
        h = self._handles.get(device)
        if h is None:
            h = self._handles[device] = _lib.Handle(cp, device)
            h.call("vdyn_set_option", _lib.VDYN_OPT_LANES_PER_ROLLOUT, self.lanes_per_rollout)
        else:
            h.set_params(cp)
        return h

    @staticmethod
    def _mu4(mu_max):
        if mu_max is None:
            return None, None
        a = np.ascontiguousarray(mu_max, dtype=np.float64)
        if a.shape != (4,):
            raise ValueError("mu_max must have 4 entries (FL, FR, RL, RR)")
        return a, a.ctypes.data_as(C.c_void_p)

    # ------------------------------------------------- reference-signature drop-ins
    def planar_model(self, state, tire_torques, mu_max, delta, p, ax_prev, ay_prev):
        """vehicle_model.py:220-425: returns
        ``[state_dot(10), vx, vy, ax, ay, outputs(18), axc, ayc]`` (:425)."""
        h, a = self._pack_scalar_call(state, tire_torques, mu_max, delta, p, ax_prev, ay_prev)
        # in: state[0:10] acc_prev[10:12] ctrl12[12:24]; out: state_dot[0:10] aux[10:14] outputs[14:32] acc[32:34]
        h.call("vdyn_planar_model_f64_host", 1, a[0], a[24], a[20], a[40], a[50], a[54], a[72])
        o = self._sc_out.copy()                                           # fresh arrays, like the reference
        return [o[0:10], o[10], o[11], o[12], o[13], o[14:32], o[32], o[33]]

    def planar_model_RK4(self, state, tire_torques, mu_max, delta, p, ax_prev, ay_prev):
        """vehicle_model.py:427-445: returns
        ``[state_update(10), x, y, yaw, U, state_dot(10), outputs(18), axc, ayc]`` (:445)."""
        h, a = self._pack_scalar_call(state, tire_torques, mu_max, delta, p, ax_prev, ay_prev)
        # out: state12[0:12] state_dot[12:22] outputs[22:40]
        h.call("vdyn_step_f64_host", 1, a[0], a[24], 12, float(self.dt), None, a[40], a[52], a[62])
        o = self._sc_out.copy()
        if not np.isfinite(o[0:12]).all():      # the reference: inf / nan + NumPy's RuntimeWarning (vehicle_model.py:284-293)
            import warnings
            warnings.warn("non-finite value encountered in planar_model_RK4 (division by a zero wheel-plane speed)",
                          RuntimeWarning, stacklevel=2)
        su = o[0:10]
        return [su, su[8], su[9], su[7], su[0], o[12:22], o[22:40], o[10], o[11]]

    _P_ATTRS = ("m", "a", "b", "Izz", "Jw", "hg", "T", "wL", "wR", "rw", "BFL", "BFR", "BRL", "BRR",
                "CFL", "CFR", "CRL", "CRR")

    def _scalar_handle(self, p):
        """Handle for the single-vehicle drop-ins; the VdynParams conversion of `p` is cached
        and redone only when one of the attributes the path reads has changed."""
        fp = tuple(getattr(p, a) for a in self._P_ATTRS)
        if getattr(self, "_scalar_fp", None) != fp:
            self._scalar_cp = params_to_c(p)
            self._scalar_key = self._scalar_cp.key()
            self._scalar_fp = fp
        h = self._handles.get(self.device)
        if h is None:
            h = self._handles[self.device] = _lib.Handle(self._scalar_cp, self.device)
            h.call("vdyn_set_option", _lib.VDYN_OPT_LANES_PER_ROLLOUT, self.lanes_per_rollout)
        h.set_params(self._scalar_cp, self._scalar_key)   # no-op unless another call changed them
        return h

    def _pack_scalar_call(self, state, tire_torques, mu_max, delta, p, ax_prev, ay_prev):
        """Single-vehicle calls run 10^4 times per simulated second (drive.py:114): inputs are
        written into, and outputs read from, two persistent buffers whose addresses are
        computed once (ndarray.ctypes costs more than the kernel launch)."""
        if len(state) != 10 or len(tire_torques) != 4 or len(mu_max) != 4 or len(delta) != 4:
            raise ValueError("planar model expects state[10], tire_torques[4], mu_max[4], delta[4]")
        if getattr(self, "_sc_in", None) is None:
            self._sc_in, self._sc_out = np.zeros(24), np.zeros(40)
            bi, bo = self._sc_in.ctypes.data, self._sc_out.ctypes.data
            # element index -> address: inputs 0..23, outputs 40..79
            self._sc_addr = {i: C.c_void_p(bi + 8 * i) for i in range(24)}
            self._sc_addr.update({40 + i: C.c_void_p(bo + 8 * i) for i in range(40)})
            self._sc_addr[20] = self._sc_addr[10]       # acc_prev sits behind the state
            self._sc_addr[24] = self._sc_addr[12]       # ctrl12 starts at element 12
        b = self._sc_in
        b[0:10] = state                                   # lists allowed (quirk Q9)
        b[10] = ax_prev
        b[11] = ay_prev
        b[12:16] = delta
        b[16:20] = tire_torques
        b[20:24] = mu_max
        h = self._scalar_handle(p)
        # vehicle_model.py:232-235 (quirk Q1): the reference stores mu_max into p.D**
        try:
            p.DFL, p.DFR, p.DRL, p.DRR = mu_max
        except AttributeError:
            pass
        return h, self._sc_addr

    # ----------------------------------------------------------------- batched API
    def step(self, states, controls, dt=None, mu_max=None, p=None, return_diag=False):
        """One RK4 step for N vehicles: ``states [12][N]``, ``controls [k][N]``
        (k = 2: delta_front, torque_all; k = 12: delta4, torque4, mu4).
        Returns ``states' [12][N]`` (and ``state_dot [10][N]``, ``outputs [18][N]``
        when ``return_diag``)."""
        be = _Backend(states)
        st = be.inp(states)
        if st.ndim != 2 or st.shape[0] != 12:
            raise ValueError("states must be [12][N]")
        n = st.shape[1]
        ct = be.inp(controls)
        if ct.ndim != 2 or ct.shape[0] not in (2, 12) or ct.shape[1] != n:
            raise ValueError("controls must be [2][N] or [12][N]")
        keep, mu4 = self._mu4(mu_max)
        so = be.out(12, n)
        sd = be.out(10, n) if return_diag else None
        ou = be.out(18, n) if return_diag else None
        self._handle(be.device_index(self.device), p).call(
            f"vdyn_step_{be.suffix}_{be.kind}", n, _vp(st), _vp(ct), int(ct.shape[0]),
            float(self.dt if dt is None else dt), mu4, _vp(so), _vp(sd), _vp(ou), *be.stream_args())
        del keep
        return (so, sd, ou) if return_diag else so

    def planar_model_batch(self, state, ctrl12, acc_prev, p=None, return_aux=True):
        """N derivative evaluations (vehicle_model.py:220-425): ``state [10][N]``,
        ``ctrl12 [12][N]``, ``acc_prev [2][N]`` -> ``state_dot [10][N]``,
        ``aux [4][N]`` (vx, vy, ax, ay), ``outputs [18][N]``, ``acc [2][N]``."""
        be = _Backend(state)
        st = be.inp(state)
        if st.ndim != 2 or st.shape[0] != 10:
            raise ValueError("state must be [10][N]")
        n = st.shape[1]
        c12 = be.inp(ctrl12, shape=(12, n))
        ap = be.inp(acc_prev, shape=(2, n))
        sd, ac = be.out(10, n), be.out(2, n)
        aux = be.out(4, n) if return_aux else None
        ou = be.out(18, n) if return_aux else None
        self._handle(be.device_index(self.device), p).call(
            f"vdyn_planar_model_{be.suffix}_{be.kind}", n, _vp(st), _vp(c12), _vp(ap), _vp(sd), _vp(aux),
            _vp(ou), _vp(ac), *be.stream_args())
        return sd, aux, ou, ac

    def rollout(self, states0, controls, dt=None, path_id=None, mu_max=None, traj_stride=0, p=None, out=None):
        """H zero-order-hold RK4 steps in one launch.

        ``states0 [12][N]``; ``controls [H][k][N]`` (per rollout) or, with
        ``path_id [N]``, a shared table ``[P][H][k]`` staged in LDS.
        Returns ``terminal [12][N]`` (and ``traj [H//traj_stride][12][N]`` when
        ``traj_stride > 0``).  ``out``: {"terminal", "traj"} -> the caller's own output buffers, written in
        place and returned (a trajectory array that lives across calls spares each call its page faults).

        fp32 only: ``states0 [22][N]`` selects the compensated state sum (include/vdyn.h,
        VDYN_OPT_STATE_ROWS) -- rows 12..21 carry the compensation terms of rows 0..9 (zeros to start
        with); the terminal state then has 22 rows too and continues the same sum when fed back."""
        be = _Backend(states0)
        s0 = be.inp(states0)
        if s0.ndim != 2 or s0.shape[0] not in (12, 22):
            raise ValueError("states0 must be [12][N] (or [22][N]: fp32 with compensation terms)")
        rows = int(s0.shape[0])
        if rows == 22 and be.suffix != "f32":
            raise ValueError("the compensated state sum ([22][N] states) is an fp32 option")
        n = s0.shape[1]
        ct = be.inp(controls)
        if ct.ndim != 3:
            raise ValueError("controls must be [H][k][N] or [P][H][k]")
        if path_id is None:
            H, k, nn = ct.shape
            if nn != n:
                raise ValueError("controls must be [H][k][N] with N matching states0")
            layout, P, pid = _lib.VDYN_CTRL_PER_ROLLOUT, 0, None
        else:
            P, H, k = ct.shape
            pid = be.inp(path_id, shape=(n,), int32=True)
            # host arrays are range-checked here; device tensors are not (that would
            # cost a device sync per call) -- the kernel clamps ids into [0, P)
            if not be.torch and n and (int(pid.min()) < 0 or int(pid.max()) >= P):
                raise ValueError("path_id out of range")
            layout = _lib.VDYN_CTRL_SHARED
        if k not in (2, 12):
            raise ValueError("k must be 2 or 12")
        if traj_stride < 0:
            raise ValueError("traj_stride must be >= 0")
        keep, mu4 = self._mu4(mu_max)
        # out: {"terminal": ..., "traj": ...} -- the caller's own output buffers, written in place and returned
        out = out or {}
        term = be.out(rows, n, reuse=out.get("terminal"))
        traj = be.out(H // traj_stride, 12, n, reuse=out.get("traj")) if traj_stride > 0 else None
        h = self._handle(be.device_index(self.device), p)
        with h.lock:        # the row count is handle state: set, launch and reset as one step (see _lib.Handle.lock)
            if rows != 12:
                h.call("vdyn_set_option", _lib.VDYN_OPT_STATE_ROWS, rows)
            try:
                h.call(f"vdyn_rollout_{be.suffix}_{be.kind}", n, int(H), _vp(s0), _vp(ct), int(k), layout,
                       _vp(pid), int(P), float(self.dt if dt is None else dt), mu4, _vp(term), _vp(traj),
                       int(traj_stride), *be.stream_args())
            finally:
                if rows != 12:
                    h.call("vdyn_set_option", _lib.VDYN_OPT_STATE_ROWS, 12)
        del keep
        return (term, traj) if traj_stride > 0 else term

    def rollout_spiral(self, states0, spiral_params, H, torque=100.0, dt=None, max_steer=None, mu_max=None,
                       traj_stride=0, p=None):
        """Lattice-driven rollouts: ``spiral_params [N][3]`` = (p1, p2, sf) per rollout -- the ``params``
        output of ``plan_lattice`` (``[E][P][3]`` is accepted and flattened: rollout r = ego r // P, path
        r % P) -- and step t steers both front wheels with
        ``clip(atan(wheelbase * kappa(min(U0 t dt, sf))), +-max_steer)``, kappa the cubic spiral of
        path_optimizer.py:149-154, U0 = states0[0].  No control array is read.  ``max_steer`` defaults to
        this model's (VehicleModel(wheelbase, max_steer, dt), drive.py:109)."""
        be = _Backend(states0)
        s0 = be.inp(states0)
        if s0.ndim != 2 or s0.shape[0] != 12:
            raise ValueError("states0 must be [12][N]")
        n = s0.shape[1]
        sp = be.inp(spiral_params)
        if sp.ndim == 3:
            sp = sp.reshape(-1, 3)
        if tuple(sp.shape) != (n, 3):
            raise ValueError("spiral_params must be [N][3] (or [E][P][3] with E * P == N)")
        sp = sp.contiguous() if be.torch else np.ascontiguousarray(sp)
        if H < 0 or traj_stride < 0:
            raise ValueError("need H >= 0 and traj_stride >= 0")
        keep, mu4 = self._mu4(mu_max)
        term = be.out(12, n)
        traj = be.out(int(H) // traj_stride, 12, n) if traj_stride > 0 else None
        self._handle(be.device_index(self.device), p).call(
            f"vdyn_rollout_spiral_{be.suffix}_{be.kind}", n, int(H), _vp(s0), _vp(sp), float(self.wheelbase),
            float(self.max_steer if max_steer is None else max_steer), float(torque),
            float(self.dt if dt is None else dt), mu4, _vp(term), _vp(traj), int(traj_stride), *be.stream_args())
        del keep
        return (term, traj) if traj_stride > 0 else term

    def rollout_fleet(self, states0, controls, classes, vehicle_id, dt=None, path_id=None, mu_max=None,
                      traj_stride=0):
        """``rollout`` for a heterogeneous fleet: ``classes`` is a sequence of VehicleParameters-like
        objects (at most 256), ``vehicle_id [N]`` the class of every rollout.  The classes' constants
        are staged through LDS; each lane keeps its class's set in registers."""
        be = _Backend(states0)
        s0 = be.inp(states0)
        if s0.ndim != 2 or s0.shape[0] != 12:
            raise ValueError("states0 must be [12][N]")
        n = s0.shape[1]
        ct = be.inp(controls)
        if ct.ndim != 3:
            raise ValueError("controls must be [H][k][N] or [P][H][k]")
        if path_id is None:
            H, k, nn = ct.shape
            if nn != n:
                raise ValueError("controls must be [H][k][N] with N matching states0")
            layout, P, pid = _lib.VDYN_CTRL_PER_ROLLOUT, 0, None
        else:
            P, H, k = ct.shape
            pid = be.inp(path_id, shape=(n,), int32=True)
            layout = _lib.VDYN_CTRL_SHARED
        if k not in (2, 12):
            raise ValueError("k must be 2 or 12")
        V = len(classes)
        if not 1 <= V <= 256:
            raise ValueError("1..256 vehicle classes")
        ctab = (VdynParams * V)(*[params_to_c(c) for c in classes])
        vid = be.inp(vehicle_id, shape=(n,), int32=True)
        if not be.torch and n and (int(vid.min()) < 0 or int(vid.max()) >= V):
            raise ValueError("vehicle_id out of range")
        keep, mu4 = self._mu4(mu_max)
        term = be.out(12, n)
        traj = be.out(H // traj_stride, 12, n) if traj_stride > 0 else None
        self._handle(be.device_index(self.device)).call(
            f"vdyn_rollout_fleet_{be.suffix}_{be.kind}", n, int(H), _vp(s0), _vp(ct), int(k), layout, _vp(pid),
            int(P), ctab, V, _vp(vid), float(self.dt if dt is None else dt), mu4, _vp(term), _vp(traj),
            int(traj_stride), *be.stream_args())
        del keep
        return (term, traj) if traj_stride > 0 else term

    def mpc_argmin(self, ego, cand, goal, dt=None, w_delta=1e-3, return_costs=False, p=None):
        """BASELINE config 5: ``ego [12][E]``, shared candidates ``cand [H][2][C]``,
        ``goal [2][E]`` -> ``(best_cost [E], best_idx [E])`` (+ ``cost [E][C]``)."""
        be = _Backend(ego)
        eg = be.inp(ego)
        if eg.ndim != 2 or eg.shape[0] != 12:
            raise ValueError("ego must be [12][E]")
        E = eg.shape[1]
        cd = be.inp(cand)
        if cd.ndim != 3 or cd.shape[1] != 2 or cd.shape[2] < 1:
            raise ValueError("cand must be [H][2][C] with C >= 1")
        H, _, Cn = cd.shape
        gl = be.inp(goal, shape=(2, E))
        bc, bi = be.out(E), be.out(E, int32=True)
        costs = be.out(E, Cn) if return_costs else None
        self._handle(be.device_index(self.device), p).call(
            f"vdyn_mpc_argmin_{be.suffix}_{be.kind}", int(E), int(Cn), int(H), _vp(eg), _vp(cd), _vp(gl),
            float(self.dt if dt is None else dt), float(w_delta), _vp(bc), _vp(bi), _vp(costs),
            *be.stream_args())
        return (bc, bi, costs) if return_costs else (bc, bi)

    # ------------------------------------------------- controllers either side of the path
    def _closed_loop_inputs(self, be, states, cstate, waypoints, wcount, path_id):
        st = be.inp(states)
        if st.ndim != 2 or st.shape[0] != 12:
            raise ValueError("states must be [12][N]")
        n = st.shape[1]
        cs = be.inp(cstate, shape=(6, n))
        wp = be.inp(waypoints)
        if wp.ndim == 2:
            wp = wp[None]
        if wp.ndim != 3 or wp.shape[2] != 2 or wp.shape[1] < 1:
            raise ValueError("waypoints must be [P][Wmax][2] (x, y)")
        P, Wmax = int(wp.shape[0]), int(wp.shape[1])
        if _is_torch_cuda(wcount):     # device tensor: not range-checked here, the kernel clamps to [1, Wmax]
            wc = be.inp(wcount, shape=(P,), int32=True)
        else:
            wc = np.full(P, Wmax, dtype=np.int32) if wcount is None else np.ascontiguousarray(wcount, dtype=np.int32)
            if wc.shape != (P,) or wc.min() < 1 or wc.max() > Wmax:
                raise ValueError("wcount must be [P] with 1 <= wcount <= Wmax")
            wc = be.inp(wc, int32=True)
        pid = be.inp(np.zeros(n, dtype=np.int32) if path_id is None else path_id, shape=(n,), int32=True)
        if not be.torch and n and (int(pid.min()) < 0 or int(pid.max()) >= P):
            raise ValueError("path_id out of range")
        return st, n, cs, wp.contiguous() if be.torch else np.ascontiguousarray(wp), P, Wmax, wc, pid

    def controller_update(self, states, cstate, waypoints, wcount=None, path_id=None, gains=None, dt=None):
        """One Stanley + PID + steering-filter update for N vehicles (drive.py:128-138).

        ``states [12][N]``; ``cstate [6][N]`` rows ``x_del, total_vel_error, prev_vel,
        target_vel, delta, torque``; ``waypoints [P][Wmax][2]``.  Returns the new
        ``cstate [6][N]`` and ``out [3][N]`` = (limited Stanley angle before the filter,
        target index, crosstrack error)."""
        be = _Backend(states)
        st, n, cs, wp, P, Wmax, wc, pid = self._closed_loop_inputs(be, states, cstate, waypoints, wcount, path_id)
        g = gains if gains is not None else _lib.default_ctrl_gains()
        cso, out = be.out(6, n), be.out(3, n)
        self._handle(be.device_index(self.device)).call(
            f"vdyn_controller_update_{be.suffix}_{be.kind}", C.byref(g), n, _vp(st), _vp(cs), _vp(wp), Wmax,
            _vp(wc), _vp(pid), P, float(self.dt if dt is None else dt), _vp(cso), _vp(out), *be.stream_args())
        return cso, out

    def closed_loop(self, states0, cstate0, waypoints, H, wcount=None, path_id=None, gains=None, dt=None,
                    ctrl_every=10, phase=0, log=False, datalog=False, out=None):
        """H sub-steps of the reference's Car.drive loop (drive.py:114-151) minus the planner:
        controllers every ``ctrl_every`` steps (zero-order hold), RK4 every step.
        Returns ``terminal [12][N]``, ``cstate [6][N]`` (and ``log [H][16][N]``: state12, delta,
        torque, target index, crosstrack error; and/or ``datalog [H][45][N]``: the reference's
        DataLog columns, drive.py:145-151 / plots.py:19-27, with t = (phase + step) * dt).
        ``out``: {"terminal", "cstate", "log", "datalog"} -> the caller's own output buffers (any subset), written in
        place and returned; worth it for the logs of a large fleet from NumPy memory (see ``_Backend.out``)."""
        be = _Backend(states0)
        st, n, cs, wp, P, Wmax, wc, pid = self._closed_loop_inputs(be, states0, cstate0, waypoints, wcount, path_id)
        if H < 0 or ctrl_every <= 0 or phase < 0:
            raise ValueError("need H >= 0, ctrl_every > 0, phase >= 0")
        g = gains if gains is not None else _lib.default_ctrl_gains()
        out = out or {}
        term, cso = be.out(12, n, reuse=out.get("terminal")), be.out(6, n, reuse=out.get("cstate"))
        lg = be.out(int(H), 16, n, reuse=out.get("log")) if log else None
        dl = be.out(int(H), 45, n, reuse=out.get("datalog")) if datalog else None
        self._handle(be.device_index(self.device)).call(
            f"vdyn_closed_loop_{be.suffix}_{be.kind}", C.byref(g), n, int(H), int(ctrl_every), int(phase),
            _vp(st), _vp(cs), _vp(wp), Wmax, _vp(wc), _vp(pid), P, float(self.dt if dt is None else dt),
            _vp(term), _vp(cso), _vp(lg), _vp(dl), *be.stream_args())
        return (term, cso) + ((lg,) if log else ()) + ((dl,) if datalog else ())

    # ------------------------------------------------- collision check + best-path selection
    def select_best_path(self, paths, obstacles, goal, circle_offsets=(-1.0, 1.0, 3.0),
                         circle_radii=(1.5, 1.5, 1.5), weight=10.0, collision_free=None, validity=None):
        """collision_checker.py:32-117 + :134-203 for E egos x P paths x L points.

        ``paths [E][P][3][L]`` (rows x, y, yaw: the reference's path lists); ``obstacles
        [M][2]`` shared or ``[E][M][2]``; ``goal [2][E]``.  ``collision_free [E][P]`` given:
        skip the check (select_best_path_index alone).  ``validity [E][P]`` (``plan_lattice``'s output):
        spirals the planner dropped (local_planner.py:312-321) are absent -- never selectable, no proximity
        penalty from them.  Returns ``collision_free [E][P]`` bool, ``best_idx [E]`` (-1 = None; an index
        into all P paths) and ``best_score [E]``.  Defaults: drive.py:25-28.
        torch CUDA tensors (e.g. ``plan_lattice``'s output) stay on the device."""
        if _is_torch_cuda(paths):
            return self._select_best_path_dev(paths, obstacles, goal, circle_offsets, circle_radii, weight,
                                              collision_free, validity)
        pa = np.ascontiguousarray(paths)
        dtype = pa.dtype if pa.dtype in (np.float32, np.float64) else np.dtype(np.float64)
        pa = pa.astype(dtype, copy=False)
        if pa.ndim != 4 or pa.shape[2] != 3:
            raise ValueError("paths must be [E][P][3][L]")
        E, P, _, L = pa.shape
        if P > 64:
            raise ValueError("at most 64 paths per ego")
        ob = np.ascontiguousarray(obstacles, dtype=dtype).reshape((-1, 2) if np.ndim(obstacles) < 3 else
                                                                  np.shape(obstacles))
        per_ego = ob.ndim == 3
        if ob.shape[-1] != 2 or (per_ego and ob.shape[0] != E):
            raise ValueError("obstacles must be [M][2] or [E][M][2]")
        M = ob.shape[-2]
        gl = np.ascontiguousarray(goal, dtype=dtype)
        if gl.shape != (2, E):
            raise ValueError("goal must be [2][E]")
        off = np.ascontiguousarray(circle_offsets, dtype=np.float64)
        rad = np.ascontiguousarray(circle_radii, dtype=np.float64)
        if off.shape != rad.shape or off.ndim != 1 or not 1 <= off.size <= 8:
            raise ValueError("1..8 circle offsets / radii")
        cin = None if collision_free is None else np.ascontiguousarray(collision_free, dtype=np.int32)
        if cin is not None and cin.shape != (E, P):
            raise ValueError("collision_free must be [E][P]")
        val = None if validity is None else np.ascontiguousarray(validity, dtype=np.int32)
        if val is not None and val.shape != (E, P):
            raise ValueError("validity must be [E][P]")
        free, bi, bs = np.empty((E, P), np.int32), np.empty(E, np.int32), np.empty(E, dtype)
        self._handle(self.device).call(
            f"vdyn_select_best_path_{_suffix(dtype)}_host", E, P, L, _vp(pa), _vp(ob), M, int(per_ego), _vp(off),
            _vp(rad), int(off.size), _vp(gl), float(weight), _vp(cin), _vp(val), _vp(free), _vp(bi), _vp(bs))
        return free.astype(bool), bi, bs

    def _select_best_path_dev(self, paths, obstacles, goal, circle_offsets, circle_radii, weight, collision_free,
                              validity=None):
        be = _Backend(paths)
        pa = be.inp(paths)
        if pa.ndim != 4 or pa.shape[2] != 3:
            raise ValueError("paths must be [E][P][3][L]")
        E, P, _, L = (int(v) for v in pa.shape)
        if P > 64:
            raise ValueError("at most 64 paths per ego")
        ob = be.inp(obstacles)
        per_ego = ob.ndim == 3
        if ob.shape[-1] != 2 or (per_ego and ob.shape[0] != E):
            raise ValueError("obstacles must be [M][2] or [E][M][2]")
        gl = be.inp(goal, shape=(2, E))
        off = np.ascontiguousarray(circle_offsets, dtype=np.float64)
        rad = np.ascontiguousarray(circle_radii, dtype=np.float64)
        if off.shape != rad.shape or off.ndim != 1 or not 1 <= off.size <= 8:
            raise ValueError("1..8 circle offsets / radii")
        cin = None if collision_free is None else be.inp(collision_free, shape=(E, P), int32=True)
        val = None if validity is None else be.inp(validity, shape=(E, P), int32=True)
        free, bi, bs = be.out(E, P, int32=True), be.out(E, int32=True), be.out(E)
        esz, base = pa.element_size(), pa.data_ptr()
        self._handle(be.device_index(self.device)).call(
            f"vdyn_select_best_path_{be.suffix}_dev", E, P, L, C.c_void_p(base), C.c_void_p(base + L * esz),
            C.c_void_p(base + 2 * L * esz), P * 3 * L, 3 * L, 1, _vp(ob), int(ob.shape[-2]), int(per_ego), _vp(off),
            _vp(rad), int(off.size), _vp(gl), float(weight), _vp(cin), _vp(val), _vp(free), _vp(bi), _vp(bs),
            *be.stream_args())
        return free.bool(), bi, bs

    def select_best_rollout(self, traj, paths_per_ego, obstacles, goal, circle_offsets=(-1.0, 1.0, 3.0),
                            circle_radii=(1.5, 1.5, 1.5), weight=10.0, validity=None):
        """The same selection fed in place by the trajectory output of ``rollout`` on the GPU:
        ``traj [L][12][N]`` (torch CUDA tensor, N = E * paths_per_ego, ego-major), obstacles
        ``[M][2]`` and ``goal [2][E]`` tensors on the same device."""
        be = _Backend(traj)
        if not be.torch or traj.ndim != 3 or traj.shape[1] != 12:
            raise ValueError("traj must be a torch CUDA tensor [L][12][N]")
        tr = be.inp(traj)
        L, _, N = tr.shape
        P = int(paths_per_ego)
        if P < 1 or P > 64 or N % P:
            raise ValueError("N must be a multiple of paths_per_ego (<= 64)")
        E = N // P
        ob = be.inp(obstacles)
        gl = be.inp(goal, shape=(2, E))
        if ob.ndim != 2 or ob.shape[1] != 2:
            raise ValueError("obstacles must be [M][2]")
        off = np.ascontiguousarray(circle_offsets, dtype=np.float64)
        rad = np.ascontiguousarray(circle_radii, dtype=np.float64)
        val = None if validity is None else be.inp(validity, shape=(E, P), int32=True)
        free, bi, bs = be.out(E, P, int32=True), be.out(E, int32=True), be.out(E)
        esz = tr.element_size()
        base = tr.data_ptr()
        row = lambda r: C.c_void_p(base + r * N * esz)
        self._handle(be.device_index(self.device)).call(
            f"vdyn_select_best_path_{be.suffix}_dev", E, P, int(L), row(8), row(9), row(7), P, 1, 12 * N,
            _vp(ob), int(ob.shape[0]), 0, _vp(off), _vp(rad), int(off.size), _vp(gl), float(weight), None, _vp(val),
            _vp(free), _vp(bi), _vp(bs), *be.stream_args())
        return free, bi, bs

    # ------------------------------------------------- lattice generation
    def plan_lattice(self, px, py, ego, goal_v, lookahead=30.0, num_paths=7, path_offset=2.0, spiral_params=None):
        """One planning cycle of LocalPlanner.MotionPlanner up to the transformed lattice
        (local_planner.py:362-368) for E egos: ``px, py [nwp]`` global path, ``ego [3][E]`` rows
        x, y, yaw.  Returns a dict: ``closest_index [E]``, ``goal_index [E]``, ``closest_len [E]``,
        ``goal_set [E][P][4]``, ``params [E][P][3]``, ``paths [E][P][3][49]``, ``validity [E][P]``,
        ``cost [E][P]``.  ``spiral_params [E][P][3]`` skips the optimiser.  Defaults: drive.py:21,24,35."""
        be = _Backend(ego)
        eg = be.inp(ego)
        if eg.ndim != 2 or eg.shape[0] != 3:
            raise ValueError("ego must be [3][E] (x, y, yaw)")
        E, P = int(eg.shape[1]), int(num_paths)
        pxx, pyy = be.inp(px), be.inp(py)
        if pxx.ndim != 1 or pxx.shape != pyy.shape or pxx.shape[0] < 2:
            raise ValueError("px, py must be 1-D of equal length >= 2")
        if P < 1:
            raise ValueError("num_paths must be >= 1")
        pin = None if spiral_params is None else be.inp(spiral_params, shape=(E, P, 3))
        ci, gi, val = be.out(E, int32=True), be.out(E, int32=True), be.out(E, P, int32=True)
        cl, gs, pr, pa, co = be.out(E), be.out(E, P, 4), be.out(E, P, 3), be.out(E, P, 3, 49), be.out(E, P)
        self._handle(be.device_index(self.device)).call(
            f"vdyn_plan_lattice_{be.suffix}_{be.kind}", E, _vp(pxx), _vp(pyy), int(pxx.shape[0]), _vp(eg),
            float(goal_v), float(lookahead), P, float(path_offset), _vp(pin), _vp(ci), _vp(gi), _vp(cl), _vp(gs),
            _vp(pr), _vp(pa), _vp(val), _vp(co), *be.stream_args())
        return dict(closest_index=ci, goal_index=gi, closest_len=cl, goal_set=gs, params=pr, paths=pa,
                    validity=val, cost=co)

    def interpolate_waypoints(self, paths, best_idx, res=0.01, Wmax=4096, out=None):
        """local_planner.py:395-419 for E egos: ``paths [E][P][3][L]``, ``best_idx [E]`` ->
        ``wp [E][Wmax][2]``, ``wcount [E]``: the tables ``closed_loop`` / ``controller_update`` take.
        An ego with ``best_idx < 0`` (no selectable path) keeps its previous table, as the reference keeps
        ``_prev_best_path`` (local_planner.py:380-384): pass last cycle's ``(wp, wcount)`` as ``out`` and they
        are updated in place; without ``out`` such an ego gets ``wcount = 0`` and an all-zero table."""
        be = _Backend(paths)
        pa = be.inp(paths)
        if pa.ndim != 4 or pa.shape[2] != 3 or pa.shape[3] < 2:
            raise ValueError("paths must be [E][P][3][L]")
        E, P, _, L = (int(v) for v in pa.shape)
        bi = be.inp(best_idx, shape=(E,), int32=True)
        if out is not None:
            wp, wc = out
            if be.torch:
                ok = _is_torch_cuda(wp) and _is_torch_cuda(wc) and wp.dtype == be.t_dtype and wp.is_contiguous() \
                    and wc.is_contiguous() and str(wc.dtype) == "torch.int32"
            else:
                ok = isinstance(wp, np.ndarray) and isinstance(wc, np.ndarray) and wp.dtype == be.np_dtype and \
                    wc.dtype == np.int32 and wp.flags.c_contiguous and wc.flags.c_contiguous
            if not ok or tuple(wp.shape) != (E, int(Wmax), 2) or tuple(wc.shape) != (E,):
                raise ValueError("out must be (wp [E][Wmax][2] of the call's dtype, wcount [E] int32), contiguous")
        elif be.torch:
            wp = be._t.zeros((E, int(Wmax), 2), dtype=be.t_dtype, device=be.device)
            wc = be._t.zeros((E,), dtype=be._t.int32, device=be.device)
        else:
            wp, wc = np.zeros((E, int(Wmax), 2), be.np_dtype), np.zeros(E, np.int32)
        self._handle(be.device_index(self.device)).call(
            f"vdyn_interpolate_waypoints_{be.suffix}_{be.kind}", E, P, L, _vp(pa), _vp(bi), float(res), int(Wmax),
            _vp(wp), _vp(wc), *be.stream_args())
        return wp, wc

    def nonfinite_lanes(self, x):
        """``x [rows][N]`` (terminal states, a trajectory slice ...) -> ``(status [N] int32, count)``:
        status 1 where any row of the lane is inf / NaN -- the batched counterpart of the RuntimeWarning
        NumPy raises in the reference when a wheel speed is zero (vehicle_model.py:284-293)."""
        be = _Backend(x)
        a = be.inp(x)
        if a.ndim != 2:
            raise ValueError("x must be [rows][N]")
        rows, n = int(a.shape[0]), int(a.shape[1])
        st = be.out(n, int32=True)
        cnt = C.c_int64(0)
        self._handle(be.device_index(self.device)).call(
            f"vdyn_nonfinite_lanes_{be.suffix}_{be.kind}", rows, n, _vp(a), _vp(st), C.byref(cnt), *be.stream_args())
        return st, int(cnt.value)

    def fastmath_eval(self, fn, x, c=0.0):
        """Device self-test hook (include/vdyn.h, vdyn_fastmath_eval_*): evaluate elementary function
        ``fn`` of the FAST step on ``x [N]`` -> ``(out0, out1)``."""
        be = _Backend(x)
        a = be.inp(x)
        if a.ndim != 1:
            raise ValueError("x must be 1-D")
        o0, o1 = be.out(a.shape[0]), be.out(a.shape[0])
        self._handle(be.device_index(self.device)).call(
            f"vdyn_fastmath_eval_{be.suffix}_{be.kind}", int(fn), int(a.shape[0]), _vp(a), float(c), _vp(o0), _vp(o1),
            *be.stream_args())
        return o0, o1

    @staticmethod
    def tire_fit(shape_factor, dtype=np.float32):
        """The FAST step's fit for one wheel with Pacejka shape factor C (include/vdyn.h,
        ``vdyn_tire_fit_f32`` / ``_f64``; host arithmetic, no device needed): ``(coef, validated)`` --
        9 float32 or 17 float64 coefficients, highest degree first -- with
        ``sin(C atan x) / x = c P(c)``, ``c = 1 / sqrt(1 + x^2)``."""
        f32 = np.dtype(dtype) == np.float32
        coef = np.zeros(9 if f32 else 17, dtype=np.float32 if f32 else np.float64)
        fn = _lib.load().vdyn_tire_fit_f32 if f32 else _lib.load().vdyn_tire_fit_f64
        return coef, fn(float(shape_factor), coef.ctypes.data_as(C.c_void_p)) == 0

    def synchronize(self, device=None):
        """Wait for the default stream of `device` (NumPy calls are already synchronous)."""
        d = self.device if device is None else device
        self._handle(d).call("vdyn_stream_synchronize", None)
