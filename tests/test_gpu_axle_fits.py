"""GPU (-m gpu): fp64 handles whose tires differ by AXLE (C_FL = C_FR != C_RL = C_RR -- the handle the reference
itself sketches, vehicle_model.py:237-242: rear B and C scaled by 0.8) through every fp64 lane kernel.

Such a handle takes the fitted tire chain with TWO coefficient sets pinned in registers (fit_mode 3 in
csrc/vdyn_kernels.hip: rollout, lattice-driven rollout, MPC argmin in both mappings, closed loop with and without
the log; the DataLog closed loop and handles with four different C read the per-wheel table from LDS) instead of the
general atan -> sine chain at half the speed.  Everything here is held to the oracle at the fp64 bar."""
import numpy as np
import pytest

from conftest import load_golden, parity

pytestmark = pytest.mark.gpu

F64_TOL = 1e-6


def _axle_vehicle(pkg, scale=0.8):
    v = pkg.VehicleParameters()
    v.CRL = v.CRR = scale * v.CFL                      # vehicle_model.py:238-239
    v.BRL = v.BRR = scale * v.BFL                      # :241-242
    return v


def _four_c_vehicle(pkg):
    v = pkg.VehicleParameters()
    v.CFR, v.CRL, v.CRR = 1.45, 1.2, 1.25
    return v


def test_axle_handles_fit_and_differ_from_the_reference_tires(pkg, gpu_vm, oracle, workloads):
    veh = _axle_vehicle(pkg)
    assert veh.CFL == veh.CFR and veh.CRL == veh.CRR and veh.CRL != veh.CFL
    assert pkg.VehicleModel.tire_fit(veh.CFL)[1] and pkg.VehicleModel.tire_fit(veh.CRL)[1]
    s0, tab, pid = workloads.config3(700, 60, np.float64)
    a = gpu_vm(1e-3, params=veh).rollout(s0, tab, path_id=pid)
    b = gpu_vm(1e-3).rollout(s0, tab, path_id=pid)
    assert np.abs(a - b).max() > 1e-6, "the rear tires must matter, or the cases below test nothing"


@pytest.mark.parametrize("k", [2, 12])
def test_rollout_kernel_axle_fits_all_layouts(pkg, gpu_vm, oracle, workloads, k):
    """Shared table through LDS, per-rollout controls, trajectory rows, k = 2 (bicycle steering) and k = 12 (four
    steering angles, four torques, four mu) against the oracle; a handle with four different C (LDS table) as well,
    and both against each other where they must agree: rear C equal -> the two paths run the same arithmetic."""
    rng = np.random.default_rng(5)
    n, H, dt = 1500, 80, 1e-3
    s0, tab, pid = workloads.config3(n, H, np.float64)
    tab[:, :, 0] *= 5.0                                # up to 0.3 rad: well into the nonlinear range of the fit
    if k == 12:
        t12 = np.empty((tab.shape[0], H, 12))
        t12[:, :, 0:2] = tab[:, :, 0:1]
        t12[:, :, 2:4] = 0.02 * rng.standard_normal((tab.shape[0], H, 2))          # rear steering: STEERED rear wheels
        t12[:, :, 4:8] = tab[:, :, 1:2] * rng.uniform(0.5, 1.0, (tab.shape[0], H, 4))
        t12[:, :, 8:12] = rng.uniform(0.7, 1.0, (tab.shape[0], H, 4))
        tab = t12
    for veh in (_axle_vehicle(pkg), _axle_vehicle(pkg, 1.15), _four_c_vehicle(pkg)):
        p = oracle.params_from(veh)
        vm = gpu_vm(dt, params=veh)
        want, wtraj = oracle.rollout(p, s0, tab, dt, path_id=pid, traj_stride=20)
        got, traj = vm.rollout(s0, tab, path_id=pid, traj_stride=20)
        assert parity(got, want, F64_TOL, "axle fits, shared table") <= 1e-9
        assert parity(traj, wtraj, F64_TOL, "axle fits, trajectory") <= 1e-9
        assert np.array_equal(vm.rollout(s0, tab, path_id=pid), got), "trajectory instance == plain instance"
        ctrl = workloads.expand_shared_controls(tab, pid)
        assert np.array_equal(vm.rollout(s0, ctrl), got), "per-rollout controls == shared table, bit for bit"
        # split horizon: the state carries everything
        half = vm.rollout(s0, tab[:, :H // 2], path_id=pid)
        assert np.array_equal(vm.rollout(half, tab[:, H // 2:], path_id=pid), got)


def test_axle_fits_slip_regimes_and_ragged_sizes(pkg, gpu_vm, oracle):
    """Locked, spinning, sideways and reversing vehicles (quirk Q4's |vx|, Q5's zero slip) on batch sizes that fill
    no wave, with the rear axle on its own set."""
    rw = 0.308309813617345
    rows = []
    for U in (8.0, 25.0):
        base = np.zeros(12)
        base[0] = U
        base[3:7] = U / rw
        for edit in (lambda s: s, lambda s: s.__setitem__(slice(3, 7), 0.0), lambda s: s.__setitem__(slice(5, 7), 0.0),
                     lambda s: s.__setitem__(slice(3, 7), 3.0 * U / rw), lambda s: s.__setitem__(1, 0.9 * U),
                     lambda s: s.__setitem__(2, 1.5),
                     lambda s: (s.__setitem__(0, -U), s.__setitem__(slice(3, 7), -U / rw))):
            st = base.copy()
            edit(st)
            rows.append(st)
    s0 = np.array(rows).T
    n, H, dt = s0.shape[1], 40, 1e-3
    ctrl = np.zeros((H, 2, n))
    ctrl[:, 0, :] = np.linspace(-0.5, 0.5, n)[None, :]
    ctrl[:, 1, :] = np.where(np.arange(n) % 2 == 0, 300.0, -800.0)[None, :]
    veh = _axle_vehicle(pkg)
    want = oracle.rollout(oracle.params_from(veh), s0, ctrl, dt)
    vm = gpu_vm(dt, params=veh)
    assert parity(vm.rollout(s0, ctrl), want, F64_TOL, "axle fits, slip regimes") <= 1e-9
    for m in (1, 3, n - 1):
        assert np.array_equal(vm.rollout(s0[:, :m].copy(), ctrl[:, :, :m].copy()), vm.rollout(s0, ctrl)[:, :m])


def test_spiral_rollout_axle_fits(pkg, gpu_vm, oracle, workloads):
    n, H, dt = 7 * 300, 120, 1e-3
    s0, sp = workloads.config3_spiral(n, H, np.float64)
    veh = _axle_vehicle(pkg)
    vm = gpu_vm(dt, params=veh)
    want, wtraj = oracle.rollout_spiral(oracle.params_from(veh), s0, sp, H, dt, traj_stride=40,
                                        nthreads=oracle.max_threads())
    got, traj = vm.rollout_spiral(s0, sp, H, traj_stride=40)
    assert parity(got, want, F64_TOL, "spiral, axle fits") <= 1e-9
    assert parity(traj, wtraj, F64_TOL, "spiral trajectory, axle fits") <= 1e-9
    assert np.array_equal(vm.rollout_spiral(s0, sp, H), got)


@pytest.mark.parametrize("E,C", [(12, 200), (517, 130)])
def test_mpc_argmin_axle_fits_both_mappings(pkg, gpu_vm, oracle, workloads, E, C):
    """A workgroup per ego (E < 512) and egos on the lanes (E >= 512), fp64: costs to 1e-9, argmin identical."""
    H, dt = 9, 2e-3
    ego, _, goal = workloads.config5(E, C, 10, np.float64)
    rng = np.random.default_rng(11)
    cand = np.empty((H, 2, C))
    cand[:, 0] = np.clip(rng.normal(0.0, 0.08, (H, C)), -0.5236, 0.5236)
    cand[:, 1] = 100.0 + rng.normal(0.0, 200.0, (H, C))
    veh = _axle_vehicle(pkg)
    vm = gpu_vm(dt, params=veh)
    bc, bi, cost = vm.mpc_argmin(ego, cand, goal, w_delta=workloads.MPC_W_DELTA, return_costs=True)
    obc, obi, ocost = oracle.mpc_argmin(oracle.params_from(veh), ego, cand, goal, dt, workloads.MPC_W_DELTA,
                                        nthreads=oracle.max_threads(), return_costs=True)
    assert np.abs(cost - ocost).max() <= 1e-9 and np.array_equal(bi, obi)
    assert np.array_equal(bi, cost.argmin(axis=1)) and np.array_equal(bc, cost.min(axis=1))
    # and the reference tires give other costs: the rear set was really used
    _, _, cost0 = gpu_vm(dt).mpc_argmin(ego, cand, goal, w_delta=workloads.MPC_W_DELTA, return_costs=True)
    assert np.abs(cost0 - cost).max() > 1e-9


def test_closed_loop_axle_fits_log_datalog_and_global_tables(pkg, gpu_vm, oracle):
    """Closed loop with the rear axle on its own set: no log (held sub-steps), the 16-row log, the DataLog (fit from
    LDS) and tables too large for LDS -- target indices exact, states to 1e-9, and the variants agree bit for bit
    where they run the same step."""
    g = load_golden("g9_closed_loop_controls.npz")
    dt = 1e-3
    veh = _axle_vehicle(pkg)
    vm = gpu_vm(dt, params=veh)
    gg = pkg._lib.default_ctrl_gains()
    for name, v in zip(("k", "k_soft", "max_steer", "lookahead", "deadband", "kp", "ki", "kd"), g["gains"]):
        setattr(gg, name, float(v))
    cp = oracle.ctrl_params(*g["gains"])
    rng = np.random.default_rng(3)
    n, P, H = 300, 3, 90
    wc = np.array([700, 650, 500], dtype=np.int32)
    wp = np.zeros((P, 700, 2))
    for p in range(P):
        wp[p, :wc[p]] = g["waypoints"][p, 400:400 + wc[p] * 3:3, :2]
    pid = rng.integers(0, P, n).astype(np.int32)
    s0 = np.tile(np.concatenate([g["state"], [0.0, 0.0]])[:, None], (1, n))
    s0[0] += rng.uniform(-3, 3, n)
    s0[3:7] = s0[0] / 0.308309813617345
    s0[7] += rng.normal(0, 0.03, n)
    s0[8] += rng.uniform(0.0, 4.0, n)
    s0[9] += rng.normal(0, 0.5, n)
    c0 = np.zeros((6, n))
    c0[2] = s0[0]
    c0[3] = 25.0
    ot, oc, olog = oracle.closed_loop(oracle.params_from(veh), cp, s0, c0, wp, wc, pid, dt, H, log=True, nthreads=8)
    term, cs, log = vm.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg, log=True)
    assert np.array_equal(log[:, 14], olog[:, 14]), "target indices must match the oracle exactly"
    assert parity(term, ot, F64_TOL, "closed loop, axle fits") <= 1e-9
    assert parity(cs[[0, 1, 2, 4, 5]], oc[[0, 1, 2, 4, 5]], F64_TOL) <= 1e-9
    t2, c2 = vm.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg)
    assert np.array_equal(t2, term) and np.array_equal(c2, cs), "held sub-steps == logged sub-steps"
    big = np.zeros((P, 9000, 2))
    big[:, :700] = wp
    t3, c3 = vm.closed_loop(s0, c0, big, H, wcount=wc, path_id=pid, gains=gg)
    assert np.array_equal(t3, term) and np.array_equal(c3, cs), "tables through L2 == tables in LDS"
    out = vm.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg, datalog=True)
    t4, dl = out[0], out[-1]
    assert parity(t4, ot, F64_TOL, "closed loop + DataLog, axle fits") <= 1e-9
    assert dl.shape[:2] == (H, 45) and parity(dl[-1, 1:11], ot[:10], F64_TOL) <= 1e-9
    # the reference tires steer differently: the rear set was really used
    t0, _ = gpu_vm(dt).closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg)
    assert np.abs(t0 - term).max() > 1e-6
    # four different C: with the DataLog the fitted chain from the LDS table, without it the general chain -- both
    # against the oracle, and against each other to rounding
    v4 = _four_c_vehicle(pkg)
    o4, _, _ = oracle.closed_loop(oracle.params_from(v4), cp, s0, c0, wp, wc, pid, dt, H, log=True, nthreads=8)
    vm4 = gpu_vm(dt, params=v4)
    t5 = vm4.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg, datalog=True)[0]
    t6 = vm4.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, gains=gg)[0]
    assert parity(t5, o4, F64_TOL, "closed loop + DataLog, four C") <= 1e-9
    assert parity(t6, o4, F64_TOL, "closed loop, four C, general chain") <= 1e-9


@pytest.mark.parametrize("k", [2, 12])
def test_axle_fits_chunked_and_too_wide_shared_tables(pkg, gpu_vm, oracle, k):
    """The per-axle instances of the two remaining control layouts: a shared table longer than one LDS chunk (several
    stagings per launch) and one too wide for LDS at all (P = 1700 paths: read through L2, layout 2 -- for k = 12 the one
    instance of the round's census with a few bytes of scratch), with trajectory rows, against the per-rollout
    expansion bit for bit and the oracle."""
    rng = np.random.default_rng(19)
    veh = _axle_vehicle(pkg)
    vm = gpu_vm(5e-4, params=veh)
    p = oracle.params_from(veh)
    for P, H, n in ((7, 900 if k == 2 else 400, 200), (1700, 6, 900)):
        tab = np.empty((P, H, k))
        if k == 2:
            tab[:, :, 0] = rng.uniform(-0.2, 0.2, (P, H))
            tab[:, :, 1] = rng.uniform(-100, 300, (P, H))
        else:
            tab[:, :, 0:2] = rng.uniform(-0.2, 0.2, (P, H, 1))
            tab[:, :, 2:4] = rng.uniform(-0.02, 0.02, (P, H, 2))
            tab[:, :, 4:8] = rng.uniform(-100, 300, (P, H, 4))
            tab[:, :, 8:12] = rng.uniform(0.5, 1.0, (P, H, 4))
        pid = rng.integers(0, P, n).astype(np.int32)
        s0 = np.zeros((12, n))
        s0[0] = rng.uniform(10, 30, n)
        s0[3:7] = s0[0] / 0.308309813617345
        a, traj = vm.rollout(s0, tab, path_id=pid, traj_stride=3)
        b = vm.rollout(s0, np.ascontiguousarray(np.transpose(tab[pid], (1, 2, 0))))
        assert np.array_equal(a, b) and np.array_equal(vm.rollout(s0, tab, path_id=pid), a)
        want, wtraj = oracle.rollout(p, s0, tab, 5e-4, path_id=pid, traj_stride=3, nthreads=4)
        assert parity(a, want, F64_TOL, f"axle fits, P={P} H={H} k={k}") <= 1e-8
        assert parity(traj, wtraj, F64_TOL, "its trajectory") <= 1e-8
