"""Child process of tests/test_host_layer_sanitizers.py: drives EVERY `_host` entry point (and the handle / peer-exchange
calls) of the library's own host layer -- csrc/vdyn_capi.hip compiled as plain C++ against tests/hipstub/ under
AddressSanitizer + UndefinedBehaviorSanitizer -- at ragged sizes, both precisions, k 2 / 12, both control layouts,
state rows 12 / 22, fleets of 1 and 256 classes, null optional outputs, allocation failures.

The stub's "kernels" read every promised input byte, write every promised output byte, and answer with an exact
function of their inputs (tests/hipstub/hip_stub.cpp), so every call is also CHECKED: a region staged at the wrong
offset, an output copied back short, a pipelined chunk uploaded over one still in use -- each changes a number here.
The stub runs queued work late and in random stream order (HIPSTUB_SEED), so a missing event dependency is a wrong
result under some seed, not a lucky pass.

No torch, no GPU, no oracle.  usage: python tests/_host_layer_driver.py <path to libvdyn_capi_asan.so>"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
L = importlib.import_module("python-motionplanning_amd._lib")

lib = C.CDLL(sys.argv[1])
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
lib.hipstub_fail_malloc_after.argtypes = [C.c_int]
lib.hipstub_live_allocations.restype = C.c_longlong
lib.hipstub_live_streams_and_events.restype = C.c_longlong
OK, ERR_ARG, ERR_OOM = 0, -1, -4
# level 2 (tests/hipstub/build_launchers.sh): the REAL launchers' host halves run, a launch is only checked for its
# geometry and counted -- nothing is computed, so content checks are off and launch counts are checked instead
LEVEL2 = os.environ.get("HIPSTUB_LEVEL") == "2"
# HIPSTUB_QUICK: only the calls whose staging copies are large enough for the worker threads (the ThreadSanitizer run)
QUICK = bool(os.environ.get("HIPSTUB_QUICK"))
if LEVEL2:
    lib.hipstub_launch_count.restype, lib.hipstub_launch_count.argtypes = C.c_longlong, [C.c_char_p]
    lib.hipstub_distinct_kernels.restype = C.c_longlong
rng = np.random.default_rng(int(os.environ.get("HIPSTUB_SEED", "0")) + 17)
checks = 0


def vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def ints(shape, lo=-3, hi=4, dtype=np.float64):
    """Small integers: sums of them are exact in float32 and float64."""
    return rng.integers(lo, hi, shape).astype(dtype)


def pattern(count, base, dtype):
    return (base + np.arange(count) % 251).astype(dtype)


def check(cond, what):
    global checks
    checks += 1
    if not cond:
        raise AssertionError(what)


def same(got, want, what):
    if LEVEL2:
        return check(got.shape == want.shape, what)
    check(got.shape == want.shape and np.array_equal(got, want), f"{what}: got {got.ravel()[:6]} want {want.ravel()[:6]}")


def create():
    p = L.VdynParams()
    lib.vdyn_params_default(C.byref(p))
    h = C.c_void_p()
    rc = lib.vdyn_create(C.byref(p), 0, C.byref(h))
    check(rc == OK, f"vdyn_create: {rc} {lib.vdyn_last_error(None)}")
    return h, p


def err(h):
    return (lib.vdyn_last_error(h) or b"").decode()


def toy_rollout(s0, ctrl, pid=None, mu=0.0, vid=None, stride=0):
    """Closed form of the stub's toy integrator (hip_stub.cpp, rollout_like)."""
    rows, n = s0.shape
    if pid is None:
        H, k, _ = ctrl.shape
        c = ctrl.astype(np.float64)
    else:
        P, H, k = ctrl.shape
        c = np.transpose(ctrl.astype(np.float64)[np.clip(pid, 0, P - 1)], (1, 2, 0))       # [H][k][n]
    w = np.arange(1, k + 1, dtype=np.float64)
    u = (c * w[None, :, None]).sum(axis=1) + mu                                               # [H][n]
    if vid is not None:
        u = u + vid[None, :]
    cum = np.cumsum(u, axis=0) if H else np.zeros((0, n))
    rw = np.array([i + 1 for i in range(12)] + [1] * (rows - 12), dtype=np.float64)[:, None]
    term = s0.astype(np.float64) + (rw * cum[-1][None, :] if H else 0.0)
    traj = None
    if stride > 0:
        ts = [t for t in range(H) if (t + 1) % stride == 0]
        traj = np.stack([s0[:12].astype(np.float64) + rw[:12] * cum[t][None, :] for t in ts]) if ts else np.zeros((0, 12, n))
    return term, traj, (u[-1] if H else None)


def run_rollout(h, sfx, dtype, n, H, k, shared, stride, rows=12, mu=None):
    s0 = ints((rows, n), dtype=dtype)
    if shared:
        P = 7
        ctrl, pid = ints((P, H, k), dtype=dtype), rng.integers(0, P, n).astype(np.int32)
    else:
        P, pid = 0, None
        ctrl = ints((H, k, n), dtype=dtype)
    term = np.full((rows, n), 777, dtype)
    traj = np.full((H // stride, 12, n), 777, dtype) if stride > 0 else None
    mu4 = (C.c_double * 4)(*mu) if mu is not None else None
    if rows != 12:
        check(lib.vdyn_set_option(h, L.VDYN_OPT_STATE_ROWS, rows) == OK, "set rows")
    rc = getattr(lib, f"vdyn_rollout_{sfx}_host")(h, n, H, vp(s0), vp(ctrl) if H else None, k, 1 if shared else 0, vp(pid), P, 1e-3,
                                                  mu4, vp(term), vp(traj), stride)
    if rows != 12:
        lib.vdyn_set_option(h, L.VDYN_OPT_STATE_ROWS, 12)
    what = f"rollout {sfx} n={n} H={H} k={k} shared={shared} stride={stride} rows={rows}"
    check(rc == OK, f"{what}: rc {rc} {err(h)}")
    wterm, wtraj, _ = toy_rollout(s0, ctrl, pid, sum(mu) if mu is not None else 0.0, None, stride)
    same(term, wterm.astype(dtype), what + " terminal")
    if stride > 0:
        same(traj, wtraj.astype(dtype), what + " trajectory")


def closed_loop_cases(h, sfx, dtype):
    # ---- closed loop / controller update: log, datalog optional
    g = L.VdynCtrlGains()
    lib.vdyn_ctrl_gains_default(C.byref(g))
    big_rows = (8 << 20) // (45 * 4)                    # vehicles x sub-steps beyond which the log download is pipelined
    for n, H, P, Wmax, log, dl, phase in ((1, 1, 1, 2, True, True, 0), (100, 10, 7, 50, False, False, 0), (65, 20, 3, 1024, True, False, 7),
                                          (65, 4, 3, 9, False, True, 10), (big_rows // 23 + 5, 23, 2, 6, True, True, 3),
                                          (big_rows // 9 + 1, 9, 1, 4, False, True, 0)):
        st, cs = ints((12, n), dtype=dtype), ints((6, n), dtype=dtype)
        wp, wc, pid = ints((P, Wmax, 2), -1, 2, dtype=dtype), np.full(P, Wmax, np.int32), rng.integers(0, P, n).astype(np.int32)
        term, cso = np.zeros((12, n), dtype), np.zeros((6, n), dtype)
        lg = np.zeros((H, 16, n), dtype) if log else None
        dlg = np.zeros((H, 45, n), dtype) if dl else None
        rc = getattr(lib, f"vdyn_closed_loop_{sfx}_host")(h, C.byref(g), n, H, 10, phase, vp(st), vp(cs), vp(wp), Wmax, vp(wc), vp(pid), P, 1e-3,
                                                          vp(term), vp(cso), vp(lg), vp(dlg))
        check(rc == OK, f"closed_loop: {rc} {err(h)}")
        # closed form of the stub's toy closed loop (hip_stub.cpp, closed_loop_body)
        wsum = wp.astype(np.float64).reshape(P, -1).sum(axis=1)[pid]
        s64, c64 = st.astype(np.float64), cs.astype(np.float64)
        wlog, wdl, last = [], [], np.full(n, -1.0)
        for t in range(H):
            u = (phase + t) + wsum
            s64 = s64 + np.arange(1, 13)[:, None] * u[None, :]
            if (phase + t) % 10 == 0:
                c64 = c64 + np.arange(1, 7)[:, None] * u[None, :]
                last = u
            if log:
                row = s64[np.arange(16) % 12] + np.arange(16)[:, None]
                row[15] = last                          # not carried between launches: -1 until ONE launch's first update
                wlog.append(row)
            if dl:
                row = s64[np.arange(45) % 12] + 100 + np.arange(45)[:, None]
                row[44] = last
                wdl.append(row)
        what = f"closed_loop {sfx} n={n} H={H} phase={phase}"
        same(term, s64.astype(dtype), what + " terminal")
        same(cso, c64.astype(dtype), what + " cstate")
        if log:
            same(lg, np.stack(wlog).astype(dtype), what + " log")
        if dl:
            same(dlg, np.stack(wdl).astype(dtype), what + " datalog")
        base = float(st.sum() + cs.sum() + wp.sum() + wc.sum() + pid.sum())
        co = np.zeros((3, n), dtype)
        rc = getattr(lib, f"vdyn_controller_update_{sfx}_host")(h, C.byref(g), n, vp(st), vp(cs), vp(wp), Wmax, vp(wc), vp(pid), P, 1e-3,
                                                                vp(cso), vp(co))
        check(rc == OK, f"controller_update: {rc} {err(h)}")
        same(co.ravel(), pattern(3 * n, base + 2, dtype), "controller_update out")


def main():
    h, p = create()
    check(lib.vdyn_abi_version() == L.VDYN_ABI_VERSION and lib.vdyn_build_id() == (b"hipstub2" if LEVEL2 else b"hipstub"), "abi / build id")
    for sfx, dtype in (("f32", np.float32), ("f64", np.float64)):
        # ---- rollout: whole-buffer staging (small), mapped staging (tiny), pipelined staging (> 8 MB of controls)
        for n, H, k, shared, stride in ((1, 1, 12, False, 0), (1, 0, 2, False, 0), (63, 5, 2, False, 1), (257, 9, 12, False, 4),
                                        (300, 12, 2, True, 0), (1000, 7, 12, True, 3), (5, 3, 2, True, 1)):
            run_rollout(h, sfx, dtype, n, H, k, shared, stride)
        run_rollout(h, sfx, dtype, 129, 6, 2, False, 2, mu=(1.0, 2.0, 3.0, 4.0))
        big = 9 << 20
        for k, stride in ((2, 0), (2, 1), (12, 7), (2, 13)):
            H = 40 if k == 2 else 29
            n = big // (H * k * np.dtype(dtype).itemsize) + 3
            run_rollout(h, sfx, dtype, n, H, k, False, stride)
        # shared table + a trajectory beyond 8 MB: the chunks' tables are gathered from the caller's [P][H][k]
        nbig = (9 << 20) // (12 * np.dtype(dtype).itemsize * 20) + 7
        run_rollout(h, sfx, dtype, nbig, 41, 2, True, 2)
        run_rollout(h, sfx, dtype, nbig // 2, 40, 12, True, 1)
        if QUICK:
            closed_loop_cases(h, sfx, dtype)
            continue
        # ---- a seeded random sweep around the staging decisions: control bytes just under / over the pipelining threshold,
        # two- and three-step horizons, strides that do not divide the horizon or exceed half of it (whole-buffer path)
        thr = 8 << 20
        for _ in range(14):
            k = int(rng.choice([2, 12]))
            H = int(rng.choice([2, 3, 5, 16, 31]))
            itemsize = np.dtype(dtype).itemsize
            n_thr = thr // (H * k * itemsize)
            n = max(1, int(n_thr + rng.integers(-3, 4)) if rng.integers(0, 2) else int(rng.integers(1, 700)))
            stride = int(rng.choice([0, 0, 1, 2, H // 2 + 1, H, H + 1]))
            run_rollout(h, sfx, dtype, n, H, k, False, stride)
        for lanes in (4, 0):                                # wheel-parallel kernel; 0 = chosen by the batch size
            check(lib.vdyn_set_option(h, L.VDYN_OPT_LANES_PER_ROLLOUT, lanes) == OK, "set lanes")
            run_rollout(h, sfx, dtype, 200, 8, 2, False, 2)
            run_rollout(h, sfx, dtype, 333, 5, 12, True, 0)
        check(lib.vdyn_set_option(h, L.VDYN_OPT_LANES_PER_ROLLOUT, 1) == OK, "set lanes")
        if dtype == np.float32:
            run_rollout(h, sfx, dtype, 70, 9, 2, False, 3, rows=22)
            run_rollout(h, sfx, dtype, big // (33 * 2 * 4) + 1, 33, 2, False, 11, rows=22)
        else:                                               # 22 rows are an fp32 option: an argument error, nothing touched
            s22 = ints((22, 8), dtype=dtype)
            lib.vdyn_set_option(h, L.VDYN_OPT_STATE_ROWS, 22)
            rc = lib.vdyn_rollout_f64_host(h, 8, 2, vp(s22), vp(ints((2, 2, 8), dtype=dtype)), 2, 0, None, 0, 1e-3, None, vp(s22.copy()), None, 0)
            lib.vdyn_set_option(h, L.VDYN_OPT_STATE_ROWS, 12)
            check(rc == ERR_ARG and "fp32" in err(h), f"22 rows in fp64: {rc} {err(h)}")
        # ---- step (H = 1 with diagnostics), k 2 / 12, optional outputs null / given
        for n, k, diag in ((1, 12, True), (1, 2, False), (77, 2, True), (300, 12, False)):
            s = ints((12, n), dtype=dtype)
            c = ints((k, n), dtype=dtype)
            out, sd, ou = np.full((12, n), 9, dtype), np.full((10, n), 9, dtype), np.full((18, n), 9, dtype)
            rc = getattr(lib, f"vdyn_step_{sfx}_host")(h, n, vp(s), vp(c), k, 1e-3, None, vp(out), vp(sd) if diag else None, vp(ou) if diag else None)
            check(rc == OK, f"step: {rc} {err(h)}")
            wterm, _, u = toy_rollout(s, c[None])
            same(out, wterm.astype(dtype), f"step {sfx} n={n} k={k}")
            if diag:
                same(sd, (np.arange(1, 11)[:, None] * u[None, :]).astype(dtype), "step state_dot")
                same(ou, (np.arange(101, 119)[:, None] * u[None, :]).astype(dtype), "step outputs")
        # ---- fleet: 1 and 256 classes
        for V, n, H, k, shared, stride in ((1, 50, 4, 2, False, 0), (256, 700, 5, 12, False, 5), (3, 64, 6, 2, True, 2)):
            classes = (L.VdynParams * V)(*[p] * V)
            s0 = ints((12, n), dtype=dtype)
            vid = rng.integers(0, V, n).astype(np.int32)
            if shared:
                P, ctrl, pid = 4, ints((4, H, k), dtype=dtype), rng.integers(0, 4, n).astype(np.int32)
            else:
                P, ctrl, pid = 0, ints((H, k, n), dtype=dtype), None
            term = np.full((12, n), 5, dtype)
            traj = np.full((H // stride, 12, n), 5, dtype) if stride else None
            rc = getattr(lib, f"vdyn_rollout_fleet_{sfx}_host")(h, n, H, vp(s0), vp(ctrl), k, 1 if shared else 0, vp(pid), P, classes, V,
                                                                vp(vid), 1e-3, None, vp(term), vp(traj), stride)
            check(rc == OK, f"fleet: {rc} {err(h)}")
            wterm, wtraj, _ = toy_rollout(s0, ctrl, pid, 0.0, vid.astype(np.float64), stride)
            same(term, wterm.astype(dtype), f"fleet {sfx} V={V}")
            if stride:
                same(traj, wtraj.astype(dtype), f"fleet {sfx} V={V} trajectory")
        # ---- planar_model: aux / outputs optional
        for n, full in ((1, True), (130, False), (4096, True)):
            st, c12, ap = ints((10, n), dtype=dtype), ints((12, n), dtype=dtype), ints((2, n), dtype=dtype)
            sd, acc = np.zeros((10, n), dtype), np.zeros((2, n), dtype)
            aux, ou = (np.zeros((4, n), dtype), np.zeros((18, n), dtype)) if full else (None, None)
            rc = getattr(lib, f"vdyn_planar_model_{sfx}_host")(h, n, vp(st), vp(c12), vp(ap), vp(sd), vp(aux), vp(ou), vp(acc))
            check(rc == OK, f"planar_model: {rc} {err(h)}")
            base = float(st.sum() + c12.sum() + ap.sum())
            same(sd.ravel(), pattern(10 * n, base, dtype), "planar_model state_dot")
            same(acc.ravel(), pattern(2 * n, base + 3, dtype), "planar_model acc")
            if full:
                same(aux.ravel(), pattern(4 * n, base + 1, dtype), "planar_model aux")
                same(ou.ravel(), pattern(18 * n, base + 2, dtype), "planar_model outputs")
        # ---- spiral rollout
        for n, H, stride in ((1, 3, 1), (333, 10, 0), (333, 10, 4)):
            s0, spx = ints((12, n), dtype=dtype), ints((n, 3), dtype=dtype)
            term = np.zeros((12, n), dtype)
            traj = np.zeros((H // stride, 12, n), dtype) if stride else None
            rc = getattr(lib, f"vdyn_rollout_spiral_{sfx}_host")(h, n, H, vp(s0), vp(spx), 3.0, 0.5, 100.0, 1e-3, None, vp(term), vp(traj), stride)
            check(rc == OK, f"spiral: {rc} {err(h)}")
            base = float(s0.sum() + spx.sum()) + 100.0
            same(term.ravel(), pattern(12 * n, base, dtype), "spiral terminal")
            if stride:
                same(traj.ravel(), pattern(traj.size, base + 1, dtype), "spiral trajectory")
        # ---- non-finite lanes: with and without the count
        for rows, n, want_count in ((12, 1, True), (12, 1000, False), (3, 65, True)):
            x = ints((rows, n), 0, 3, dtype=dtype)
            status = np.zeros(n, np.int32)
            cnt = C.c_int64(-1)
            rc = getattr(lib, f"vdyn_nonfinite_lanes_{sfx}_host")(h, rows, n, vp(x), vp(status), C.byref(cnt) if want_count else None)
            check(rc == OK, f"nonfinite: {rc} {err(h)}")
            same(status, pattern(n, 0, np.int32), "nonfinite status")
            if want_count and not LEVEL2:
                check(cnt.value == int(x.sum()) & 0xffff, f"nonfinite count {cnt.value}")
        # ---- fastmath eval, one / two outputs
        for n, two in ((1, True), (999, False)):
            x = ints(n, dtype=dtype)
            o0, o1 = np.zeros(n, dtype), (np.zeros(n, dtype) if two else None)
            rc = getattr(lib, f"vdyn_fastmath_eval_{sfx}_host")(h, 0, n, vp(x), 2.0, vp(o0), vp(o1))
            check(rc == OK, f"fastmath: {rc} {err(h)}")
            same(o0, pattern(n, float(x.sum()) + 2.0, dtype), "fastmath out0")
            if two:
                same(o1, pattern(n, float(x.sum()) + 3.0, dtype), "fastmath out1")
        # ---- MPC argmin: cost_all optional
        for E, Cn, H, want_all in ((1, 1, 1, True), (33, 17, 5, False), (64, 512, 10, True)):
            ego, cand, goal = ints((12, E), dtype=dtype), ints((H, 2, Cn), dtype=dtype), ints((2, E), dtype=dtype)
            bc, bi = np.zeros(E, dtype), np.full(E, -1, np.int32)
            ca = np.zeros((E, Cn), dtype) if want_all else None
            rc = getattr(lib, f"vdyn_mpc_argmin_{sfx}_host")(h, E, Cn, H, vp(ego), vp(cand), vp(goal), 2e-3, 0.0, vp(bc), vp(bi), vp(ca))
            check(rc == OK, f"mpc: {rc} {err(h)}")
            base = float(ego.sum() + cand.sum() + goal.sum())
            same(bc, pattern(E, base, dtype), "mpc best_cost")
            same(bi, pattern(E, 0, np.int32), "mpc best_idx")
            if want_all:
                same(ca.ravel(), pattern(E * Cn, base + 1, dtype), "mpc cost_all")
        closed_loop_cases(h, sfx, dtype)
        # ---- select best path: obstacles shared / per ego / none; collision_in, validity optional
        off, rad = (C.c_double * 3)(-1.0, 1.0, 3.0), (C.c_double * 3)(1.5, 1.5, 1.5)
        for E, P, Lp, M, per_ego, cin, val in ((1, 1, 1, 0, 0, False, False), (10, 7, 49, 106, 0, False, True), (5, 64, 3, 4, 1, True, False)):
            paths, goal = ints((E, P, 3, Lp), dtype=dtype), ints((2, E), dtype=dtype)
            obst = ints(((E if per_ego else 1), M, 2), dtype=dtype) if M else None
            ci = rng.integers(0, 2, (E, P)).astype(np.int32) if cin else None
            va = rng.integers(0, 2, (E, P)).astype(np.int32) if val else None
            cf, bi, bs = np.full((E, P), -1, np.int32), np.full(E, -1, np.int32), np.zeros(E, dtype)
            rc = getattr(lib, f"vdyn_select_best_path_{sfx}_host")(h, E, P, Lp, vp(paths), vp(obst), M, per_ego, off, rad, 3, vp(goal), 10.0,
                                                                   vp(ci), vp(va), vp(cf), vp(bi), vp(bs))
            check(rc == OK, f"select: {rc} {err(h)}")
            base = 7.5 + 10.0 + float(paths.sum() + goal.sum()) + (float(obst.sum()) if M else 0.0) + (float(ci.sum()) if cin else 0.0) + \
                (float(va.sum()) if val else 0.0)
            same(bs, pattern(E, base, dtype), f"select best_score E={E}")
            same(cf.ravel(), pattern(E * P, 0, np.int32), "select collision_free")
        # ---- lattice: closest_len, params_in optional;  interpolate: IN / OUT tables
        for E, P, nwp, clen, pin in ((1, 1, 2, True, False), (9, 7, 400, False, True), (40, 7, 1000, True, True)):
            px, py, ego = ints(nwp, dtype=dtype), ints(nwp, dtype=dtype), ints((3, E), dtype=dtype)
            params_in = ints((E, P, 3), dtype=dtype) if pin else None
            ci, gi = np.zeros(E, np.int32), np.zeros(E, np.int32)
            cl = np.zeros(E, dtype) if clen else None
            gs, pr, pa = np.zeros((E, P, 4), dtype), np.zeros((E, P, 3), dtype), np.zeros((E, P, 3, 49), dtype)
            va, co = np.zeros((E, P), np.int32), np.zeros((E, P), dtype)
            rc = getattr(lib, f"vdyn_plan_lattice_{sfx}_host")(h, E, vp(px), vp(py), nwp, vp(ego), 25.0, 30.0, P, 2.0, vp(params_in), vp(ci), vp(gi),
                                                               vp(cl), vp(gs), vp(pr), vp(pa), vp(va), vp(co))
            check(rc == OK, f"lattice: {rc} {err(h)}")
            base = float(px.sum() + py.sum() + ego.sum()) + (float(params_in.sum()) if pin else 0.0) + 57.0
            same(pa.ravel(), pattern(pa.size, base + 3, dtype), "lattice paths")
            same(co.ravel(), pattern(co.size, base + 4, dtype), "lattice cost")
            same(gi, pattern(E, 1, np.int32), "lattice goal_idx")
            if clen:
                same(cl, pattern(E, base, dtype), "lattice closest_len")
            Wmax = 30
            best = rng.integers(-1, P, E).astype(np.int32)
            wp_out, wcount = ints((E, Wmax, 2), dtype=dtype), np.full(E, 5, np.int32)
            keep = wp_out.copy()
            rc = getattr(lib, f"vdyn_interpolate_waypoints_{sfx}_host")(h, E, P, 49, vp(pa), vp(best), 0.5, Wmax, vp(wp_out), vp(wcount))
            check(rc == OK, f"interpolate: {rc} {err(h)}")
            b2 = float(pa.astype(np.float64).sum()) + 0.5
            for e in range(E):
                if LEVEL2:
                    continue
                if best[e] < 0:
                    check(np.array_equal(wp_out[e], keep[e]) and wcount[e] == 5, "interpolate: an ego without a path keeps its table")
                else:
                    same(wp_out[e].ravel(), pattern(Wmax * 2, b2 + e, dtype), "interpolate table")
                    check(wcount[e] == Wmax, "interpolate count")

    if QUICK:
        lib.vdyn_destroy(h)
        check(lib.hipstub_live_allocations() == 0 and lib.hipstub_live_streams_and_events() == 0, "leaks")
        print(f"host layer driver (quick): {checks} checks passed")
        return
    # ---- other tire sets: one C per axle, four different C, a shape factor no fit covers, a negative stiffness -- each takes
    # another way through the launchers (fit cache, per-wheel fit tables, the general chain); vdyn_set_params on a live handle
    def tires(**kw):
        q = L.VdynParams()
        lib.vdyn_params_default(C.byref(q))
        for i, c in enumerate(kw.get("C", ())):
            q.C[i] = c
        for i, b in enumerate(kw.get("B", ())):
            q.B[i] = b
        return q
    sets = [tires(C=(1.5047, 1.5047, 1.3, 1.3)), tires(C=(1.5, 1.45, 1.3, 1.25)), tires(C=(3.1,) * 4), tires(B=(-1.0, 20.0, 20.0, 20.0)),
            tires(C=(0.7,) * 4)]
    ht = C.c_void_p()
    check(lib.vdyn_create(C.byref(sets[0]), 0, C.byref(ht)) == OK, "create with other tires")
    for q in sets:
        check(lib.vdyn_set_params(ht, C.byref(q)) == OK, f"set_params: {err(ht)}")
        for sfx, dtype in (("f32", np.float32), ("f64", np.float64)):
            for lanes in (1, 4):
                lib.vdyn_set_option(ht, L.VDYN_OPT_LANES_PER_ROLLOUT, lanes)
                run_rollout(ht, sfx, dtype, 130, 6, 2, False, 3)
                run_rollout(ht, sfx, dtype, 70, 4, 12, True, 0)
            lib.vdyn_set_option(ht, L.VDYN_OPT_LANES_PER_ROLLOUT, 1)
            n = 40
            st, cs = ints((12, n), dtype=dtype), ints((6, n), dtype=dtype)
            wp, wc, pid = ints((2, 20, 2), dtype=dtype), np.full(2, 20, np.int32), rng.integers(0, 2, n).astype(np.int32)
            term, cso, dlg = np.zeros((12, n), dtype), np.zeros((6, n), dtype), np.zeros((5, 45, n), dtype)
            g2 = L.VdynCtrlGains()
            lib.vdyn_ctrl_gains_default(C.byref(g2))
            check(getattr(lib, f"vdyn_closed_loop_{sfx}_host")(ht, C.byref(g2), n, 5, 10, 0, vp(st), vp(cs), vp(wp), 20, vp(wc), vp(pid), 2, 1e-3,
                                                               vp(term), vp(cso), None, vp(dlg)) == OK, f"closed_loop, other tires: {err(ht)}")
            ego, cand, goal = ints((12, 9), dtype=dtype), ints((4, 2, 33), dtype=dtype), ints((2, 9), dtype=dtype)
            bc, bi = np.zeros(9, dtype), np.zeros(9, np.int32)
            check(getattr(lib, f"vdyn_mpc_argmin_{sfx}_host")(ht, 9, 33, 4, vp(ego), vp(cand), vp(goal), 2e-3, 1e-3, vp(bc), vp(bi), None) == OK,
                  f"mpc, other tires: {err(ht)}")
            s0, spx, t12 = ints((12, n), dtype=dtype), ints((n, 3), dtype=dtype), np.zeros((12, n), dtype)
            check(getattr(lib, f"vdyn_rollout_spiral_{sfx}_host")(ht, n, 5, vp(s0), vp(spx), 3.0, 0.5, 100.0, 1e-3, None, vp(t12), None, 0) == OK,
                  f"spiral, other tires: {err(ht)}")
            classes = (L.VdynParams * len(sets))(*sets)
            vid = rng.integers(0, len(sets), n).astype(np.int32)
            ctrl = ints((3, 2, n), dtype=dtype)
            check(getattr(lib, f"vdyn_rollout_fleet_{sfx}_host")(ht, n, 3, vp(s0), vp(ctrl), 2, 0, None, 0, classes, len(sets), vp(vid), 1e-3, None,
                                                                 vp(t12), None, 0) == OK, f"fleet of mixed tires: {err(ht)}")
    bad = tires()
    bad.m = float("nan")
    check(lib.vdyn_set_params(ht, C.byref(bad)) == ERR_ARG and "finite" in err(ht), "non-finite parameter")
    lib.vdyn_destroy(ht)

    # ---- argument errors come back as codes + messages, never as a crash; sizes of zero are no-ops
    check(lib.vdyn_rollout_f32_host(h, -1, 5, None, None, 2, 0, None, 0, 1e-3, None, None, None, 0) == ERR_ARG, "n < 0")
    check(lib.vdyn_rollout_f32_host(h, 0, 5, None, None, 2, 0, None, 0, 1e-3, None, None, None, 0) == OK, "n == 0")
    check(lib.vdyn_rollout_f32_host(h, 4, 5, None, None, 2, 0, None, 0, 1e-3, None, None, None, 0) == ERR_ARG and "null" in err(h), "null buffers")
    check(lib.vdyn_rollout_f64_host(h, 4, 5, None, None, 3, 0, None, 0, 1e-3, None, None, None, 0) == ERR_ARG, "k = 3")
    check(lib.vdyn_rollout_f64_host(None, 4, 5, None, None, 2, 0, None, 0, 1e-3, None, None, None, 0) == ERR_ARG, "null handle")
    check(lib.vdyn_set_option(h, 99, 1) == ERR_ARG and lib.vdyn_set_option(h, L.VDYN_OPT_STATE_ROWS, 13) == ERR_ARG, "options")
    coef32, coef64 = np.zeros(9, np.float32), np.zeros(17, np.float64)
    check(lib.vdyn_tire_fit_f32(1.5, vp(coef32)) == OK and lib.vdyn_tire_fit_f64(3.5, vp(coef64)) == ERR_ARG, "tire fit codes")
    check(lib.vdyn_tire_fit_f32(1.5, None) == ERR_ARG, "tire fit null")

    # ---- allocation failures: an error code, the handle stays usable, nothing leaks
    h2, _ = create()
    lib.hipstub_fail_malloc_after(0)
    s0, c = ints((12, 50), dtype=np.float32), ints((4, 2, 50), dtype=np.float32)
    t = np.zeros((12, 50), np.float32)
    check(lib.vdyn_rollout_f32_host(h2, 50, 4, vp(s0), vp(c), 2, 0, None, 0, 1e-3, None, vp(t), None, 0) == ERR_OOM, f"OOM staging: {err(h2)}")
    check(lib.vdyn_rollout_f32_host(h2, 50, 4, vp(s0), vp(c), 2, 0, None, 0, 1e-3, None, vp(t), None, 0) == OK, "after OOM")
    same(t, toy_rollout(s0, c)[0].astype(np.float32), "rollout after an allocation failure")
    lib.hipstub_fail_malloc_after(0)
    bc, bi = np.zeros(8, np.float32), np.zeros(8, np.int32)
    ego, cand, goal = ints((12, 8), dtype=np.float32), ints((3, 2, 600), dtype=np.float32), ints((2, 8), dtype=np.float32)
    rc = lib.vdyn_mpc_argmin_f32_host(h2, 8, 600, 3, vp(ego), vp(cand), vp(goal), 2e-3, 0.0, vp(bc), vp(bi), None)
    check(rc == ERR_OOM, f"OOM in a later allocation of the same call: {rc} {err(h2)}")
    check(lib.vdyn_mpc_argmin_f32_host(h2, 8, 600, 3, vp(ego), vp(cand), vp(goal), 2e-3, 0.0, vp(bc), vp(bi), None) == OK, "mpc after OOM")
    lib.vdyn_destroy(h2)

    # ---- peer exchange: EIGHT slot buffers, seven destinations per push through seven copy streams, fences, waits
    world, rank, block = 8, 3, 12 * 1000 * 4
    owners = [create()[0] for _ in range(world)]
    own, ipc = [C.c_void_p() for _ in range(world)], [L.VdynIpcHandle() for _ in range(world)]
    for r in range(world):
        check(lib.vdyn_xchg_alloc(owners[r], world * block, C.byref(own[r]), C.byref(ipc[r])) == OK, "xchg_alloc")
    peers = []
    for r in range(world):
        if r != rank:
            q = C.c_void_p()
            check(lib.vdyn_xchg_open(owners[rank], C.byref(ipc[r]), C.byref(q)) == OK, "xchg_open")
            peers.append(q)
    dst = (C.c_void_p * 7)(*[q.value for q in peers])
    src_dev, src_ipc = C.c_void_p(), L.VdynIpcHandle()
    check(lib.vdyn_xchg_alloc(owners[rank], 16 * block, C.byref(src_dev), C.byref(src_ipc)) == OK, "source blocks")
    stream = C.c_void_p(None)
    payload = rng.integers(0, 250, (16, block), dtype=np.uint8)
    C.memmove(src_dev.value, payload.ctypes.data, payload.nbytes)          # "device" memory is host memory in the stub
    for i in range(16):                                                    # sixteen back-to-back pushes, a fence every fourth
        check(lib.vdyn_xchg_push(owners[rank], dst, 7, rank * block, C.c_void_p(src_dev.value + i * block), block, stream) == OK, "push")
        if i % 4 == 3:
            check(lib.vdyn_xchg_fence(owners[rank], stream) == OK, "fence")
    check(lib.vdyn_xchg_wait(owners[rank]) == OK and lib.vdyn_stream_synchronize(owners[rank], stream) == OK, "wait")
    for q in peers:                                                        # the LAST block landed last, in every peer's slot `rank`
        got = np.frombuffer((C.c_ubyte * block).from_address(q.value + rank * block), dtype=np.uint8)
        check(np.array_equal(got, payload[15]), "peer slot content after sixteen pushes")
    check(lib.vdyn_xchg_push(owners[rank], dst, 65, 0, src_dev, 1, stream) == ERR_ARG, "too many destinations")
    for q in peers:
        check(lib.vdyn_xchg_close(owners[rank], q) == OK, "xchg_close")
    check(lib.vdyn_xchg_free(owners[rank], src_dev) == OK, "xchg_free")
    for r in range(world):
        check(lib.vdyn_xchg_free(owners[r], own[r]) == OK, "xchg_free")
        lib.vdyn_destroy(owners[r])

    lib.vdyn_destroy(h)
    check(lib.hipstub_live_allocations() == 0, f"{lib.hipstub_live_allocations()} device / pinned allocations outlive their handles")
    check(lib.hipstub_live_streams_and_events() == 0, f"{lib.hipstub_live_streams_and_events()} streams / events outlive their handles")
    if LEVEL2:
        fams = ("rollout_kernel", "rollout_quad_kernel", "rollout_fleet_kernel", "rollout_spiral_kernel", "planar_model_kernel",
                "mpc_", "closed_loop_kernel", "controller_kernel", "select_best_path_kernel", "lattice", "interpolate",
                "nonfinite", "fastmath")
        counts = {f: lib.hipstub_launch_count(f.encode()) for f in fams}
        check(all(v > 0 for v in counts.values()), f"kernel families never launched: {[f for f, v in counts.items() if not v]}")
        print(f"level 2: {lib.hipstub_launch_count(None)} launches of {lib.hipstub_distinct_kernels()} distinct kernel instances, "
              f"every one within its launch limits; per family: {counts}")
    print(f"host layer driver: {checks} checks passed (seed {os.environ.get('HIPSTUB_SEED', '0')}, "
          f"copy threads {os.environ.get('VDYN_COPY_THREADS', 'default')})")


if __name__ == "__main__":
    main()
