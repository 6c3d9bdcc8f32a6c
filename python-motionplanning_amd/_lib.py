"""ctypes binding of ``libvdyn_hip.so`` -- the C ABI of ``include/vdyn.h``.

There is deliberately no fallback: if the HIP library is missing or a call
fails, this raises.  Nothing here (or anywhere in the package) touches
``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

from ._build import LIB_PATH

VDYN_ABI_VERSION = 2
VDYN_OK, VDYN_ERR_ARG, VDYN_ERR_HIP, VDYN_ERR_NODEV, VDYN_ERR_OOM = 0, -1, -2, -3, -4
VDYN_CTRL_PER_ROLLOUT, VDYN_CTRL_SHARED = 0, 1
VDYN_OPT_LANES_PER_ROLLOUT = 1
VDYN_OPT_STATE_ROWS = 2


class VdynError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"vdyn error {code}: {msg}")
        self.code = code


class VdynParams(C.Structure):
    """Mirror of ``struct VdynParams`` (include/vdyn.h)."""
    _fields_ = [(n, C.c_double) for n in
                ("m", "a", "b", "Izz", "Jw", "hg", "T", "wL", "wR", "rw", "g")] + \
               [("B", C.c_double * 4), ("C", C.c_double * 4)]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_[:11]) + tuple(self.B) + tuple(self.C)


class VdynCtrlGains(C.Structure):
    """Mirror of ``struct VdynCtrlGains`` (include/vdyn.h)."""
    _fields_ = [(n, C.c_double) for n in
                ("k", "k_soft", "max_steer", "lookahead", "deadband", "kp", "ki", "kd", "filter_gain")]


_vp = C.c_void_p
_i32, _i64, _dbl, _int = C.c_int32, C.c_int64, C.c_double, C.c_int

# name -> (restype, argtypes); one entry per symbol include/vdyn.h declares
SIGNATURES = {
    "vdyn_abi_version": (_int, []),
    "vdyn_build_id": (C.c_char_p, []),
    "vdyn_device_count": (_int, []),
    "vdyn_params_default": (None, [C.POINTER(VdynParams)]),
    "vdyn_create": (_int, [C.POINTER(VdynParams), _int, C.POINTER(_vp)]),
    "vdyn_set_params": (_int, [_vp, C.POINTER(VdynParams)]),
    "vdyn_destroy": (None, [_vp]),
    "vdyn_last_error": (C.c_char_p, [_vp]),
    "vdyn_stream_synchronize": (_int, [_vp, _vp]),
    "vdyn_set_option": (_int, [_vp, _int, _int]),
    "vdyn_ctrl_gains_default": (None, [C.POINTER(VdynCtrlGains)]),
}
for _s in ("f32", "f64"):
    SIGNATURES[f"vdyn_planar_model_{_s}_dev"] = (_int, [_vp, _i64] + [_vp] * 7 + [_vp])
    SIGNATURES[f"vdyn_planar_model_{_s}_host"] = (_int, [_vp, _i64] + [_vp] * 7)
    SIGNATURES[f"vdyn_step_{_s}_dev"] = (_int, [_vp, _i64, _vp, _vp, _int, _dbl, _vp, _vp, _vp, _vp, _vp])
    SIGNATURES[f"vdyn_step_{_s}_host"] = (_int, [_vp, _i64, _vp, _vp, _int, _dbl, _vp, _vp, _vp, _vp])
    SIGNATURES[f"vdyn_rollout_{_s}_dev"] = (_int, [_vp, _i64, _i32, _vp, _vp, _int, _int, _vp, _i32, _dbl,
                                                   _vp, _vp, _vp, _i32, _vp])
    SIGNATURES[f"vdyn_rollout_{_s}_host"] = (_int, [_vp, _i64, _i32, _vp, _vp, _int, _int, _vp, _i32, _dbl,
                                                    _vp, _vp, _vp, _i32])
    SIGNATURES[f"vdyn_mpc_argmin_{_s}_dev"] = (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _dbl, _dbl,
                                                      _vp, _vp, _vp, _vp])
    SIGNATURES[f"vdyn_mpc_argmin_{_s}_host"] = (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _dbl, _dbl,
                                                       _vp, _vp, _vp])

for _s in ("f32", "f64"):
    _rs = [_vp, _i64, _i32, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _vp, _vp, _vp, _i32]
    SIGNATURES[f"vdyn_rollout_spiral_{_s}_dev"] = (_int, _rs + [_vp])
    SIGNATURES[f"vdyn_rollout_spiral_{_s}_host"] = (_int, _rs)

for _s in ("f32", "f64"):
    SIGNATURES[f"vdyn_nonfinite_lanes_{_s}_dev"] = (_int, [_vp, _i32, _i64, _vp, _vp, C.POINTER(_i64), _vp])
    SIGNATURES[f"vdyn_nonfinite_lanes_{_s}_host"] = (_int, [_vp, _i32, _i64, _vp, _vp, C.POINTER(_i64)])
    SIGNATURES[f"vdyn_fastmath_eval_{_s}_dev"] = (_int, [_vp, _i32, _i64, _vp, _dbl, _vp, _vp, _vp])
    SIGNATURES[f"vdyn_fastmath_eval_{_s}_host"] = (_int, [_vp, _i32, _i64, _vp, _dbl, _vp, _vp])
SIGNATURES["vdyn_tire_fit_f32"] = (_int, [_dbl, _vp])
SIGNATURES["vdyn_tire_fit_f64"] = (_int, [_dbl, _vp])

_gp = C.POINTER(VdynCtrlGains)
for _s in ("f32", "f64"):
    _cu = [_vp, _gp, _i64, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _dbl, _vp, _vp]
    SIGNATURES[f"vdyn_controller_update_{_s}_dev"] = (_int, _cu + [_vp])
    SIGNATURES[f"vdyn_controller_update_{_s}_host"] = (_int, _cu)
    _cl = [_vp, _gp, _i64, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _dbl, _vp, _vp, _vp, _vp]
    SIGNATURES[f"vdyn_closed_loop_{_s}_dev"] = (_int, _cl + [_vp])
    SIGNATURES[f"vdyn_closed_loop_{_s}_host"] = (_int, _cl)

for _s in ("f32", "f64"):
    SIGNATURES[f"vdyn_select_best_path_{_s}_dev"] = (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp, _i64, _i64, _i64,
                                                            _vp, _i32, _i32, _vp, _vp, _i32, _vp, _dbl, _vp, _vp, _vp,
                                                            _vp, _vp, _vp])
    SIGNATURES[f"vdyn_select_best_path_{_s}_host"] = (_int, [_vp, _i32, _i32, _i32, _vp, _vp, _i32, _i32, _vp,
                                                             _vp, _i32, _vp, _dbl, _vp, _vp, _vp, _vp, _vp])

for _s in ("f32", "f64"):
    _pl = [_vp, _i32, _vp, _vp, _i32, _vp, _dbl, _dbl, _i32, _dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]
    SIGNATURES[f"vdyn_plan_lattice_{_s}_dev"] = (_int, _pl + [_vp])
    SIGNATURES[f"vdyn_plan_lattice_{_s}_host"] = (_int, _pl)
    _iw = [_vp, _i32, _i32, _i32, _vp, _vp, _dbl, _i32, _vp, _vp]
    SIGNATURES[f"vdyn_interpolate_waypoints_{_s}_dev"] = (_int, _iw + [_vp])
    SIGNATURES[f"vdyn_interpolate_waypoints_{_s}_host"] = (_int, _iw)

for _s in ("f32", "f64"):
    _rf = [_vp, _i64, _i32, _vp, _vp, _int, _int, _vp, _i32, _vp, _i32, _vp, _dbl, _vp, _vp, _vp, _i32]
    SIGNATURES[f"vdyn_rollout_fleet_{_s}_dev"] = (_int, _rf + [_vp])
    SIGNATURES[f"vdyn_rollout_fleet_{_s}_host"] = (_int, _rf)


class VdynIpcHandle(C.Structure):
    """Mirror of ``struct VdynIpcHandle`` (include/vdyn.h): an exported device buffer, 64 opaque bytes."""
    _fields_ = [("bytes", C.c_ubyte * 64)]


_u64 = C.c_uint64
SIGNATURES.update({
    "vdyn_xchg_alloc": (_int, [_vp, _u64, C.POINTER(_vp), C.POINTER(VdynIpcHandle)]),
    "vdyn_xchg_free": (_int, [_vp, _vp]),
    "vdyn_xchg_open": (_int, [_vp, C.POINTER(VdynIpcHandle), C.POINTER(_vp)]),
    "vdyn_xchg_close": (_int, [_vp, _vp]),
    "vdyn_xchg_push": (_int, [_vp, C.POINTER(_vp), _i32, _u64, _vp, _u64, _vp]),
    "vdyn_xchg_fence": (_int, [_vp, _vp]),
    "vdyn_xchg_wait": (_int, [_vp]),
})

_lib = None


def load():
    """dlopen libvdyn_hip.so and bind every entry point; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VdynError(VDYN_ERR_NODEV,
                        f"{LIB_PATH} is not built: run __graft_entry__.build() "
                        "(there is no CPU fallback for this path)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 /
    # libhsa-runtime64, and a second HSA runtime in the same process finds no
    # device.  Importing torch first makes the loader resolve our DT_NEEDED
    # libamdhip64.so.7 to the copy torch already mapped (same SONAME), so device
    # pointers and streams are shared with torch.  Without torch installed the
    # library binds to /opt/rocm's runtime instead.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    if lib.vdyn_abi_version() != VDYN_ABI_VERSION:
        raise VdynError(VDYN_ERR_ARG, "libvdyn_hip.so ABI version mismatch")
    _lib = lib
    return lib


def build_id():
    """Identity of the loaded code objects (include/vdyn.h, vdyn_build_id)."""
    return load().vdyn_build_id().decode()


def default_params():
    p = VdynParams()
    load().vdyn_params_default(C.byref(p))
    return p


def default_ctrl_gains():
    g = VdynCtrlGains()
    load().vdyn_ctrl_gains_default(C.byref(g))
    return g


class Handle:
    """RAII wrapper of a ``VdynHandle*``."""

    def __init__(self, params: VdynParams, device: int = 0):
        self._lib = load()
        self._h = _vp()
        rc = self._lib.vdyn_create(C.byref(params), int(device), C.byref(self._h))
        if rc != VDYN_OK:
            msg = self._lib.vdyn_last_error(None)
            raise VdynError(rc, msg.decode() if msg else "vdyn_create failed")
        self.device = int(device)
        self._key = params.key()
        # handle-wide state (vdyn_set_option) is set, used by a launch and reset under this lock: two threads
        # sharing a VehicleModel must not see each other's VDYN_OPT_STATE_ROWS (a 12-row call launched while the
        # option says 22 would read and write 22 rows of 12-row buffers).  The C handle itself stays single-threaded
        # (include/vdyn.h); this only keeps the Python shim's own set / launch / reset sequences apart.
        import threading
        self.lock = threading.RLock()

    def set_params(self, params: VdynParams, key=None):
        key = params.key() if key is None else key
        if key != self._key:
            self.check(self._lib.vdyn_set_params(self._h, C.byref(params)))
            self._key = key

    def check(self, rc):
        if rc != VDYN_OK:
            msg = self._lib.vdyn_last_error(self._h)
            raise VdynError(rc, msg.decode() if msg else "?")

    def call(self, name, *args):
        self.check(getattr(self._lib, name)(self._h, *args))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.vdyn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
