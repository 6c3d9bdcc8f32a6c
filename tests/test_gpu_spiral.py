"""GPU (-m gpu): lattice-driven rollouts (vdyn_rollout_spiral_*): every rollout steers along the cubic
spiral the conformal-lattice planner gave it, delta_t = clip(atan(L kappa(min(U0 t dt, sf))), +-max_steer)
(path_optimizer.py:98-103,148-154; SURVEY.md section 8d config 3, "realistic alternative").
Pinned by G12 (spiral parameters from the reference's PathOptimizer, steering computed from them, terminal
states from the reference's planar_model_RK4) and by the oracle at BASELINE size."""
import numpy as np
import pytest

from conftest import load_golden, parity

pytestmark = pytest.mark.gpu

F64_TOL, F32_TOL = 1e-6, 1e-3


def test_g12_spiral_rollouts_vs_reference(gpu_vm):
    g = load_golden("g12_spiral_rollouts.npz")
    H, dt, tq = g["delta"].shape[0], float(g["dt"]), float(g["torque"])
    vm = gpu_vm(dt)
    term, traj = vm.rollout_spiral(g["state0"], g["params"], H, torque=tq, traj_stride=20)
    assert parity(term, g["terminal"], F64_TOL, "G12 fp64") <= 1e-9
    assert parity(traj, g["every20"], F64_TOL, "G12 fp64 every 20") <= 1e-9
    t32 = vm.rollout_spiral(g["state0"].astype(np.float32), g["params"].astype(np.float32), H, torque=tq)
    e = parity(t32, g["terminal"], F32_TOL, "G12 fp32")
    print(f"\n  G12: fp32 row-relative err {e:.2e}")
    # the table-driven kernel fed the golden steering sequence lands on the same states
    ctrl = np.ascontiguousarray(np.stack([g["delta"], np.full_like(g["delta"], tq)], axis=1))
    assert parity(vm.rollout(g["state0"], ctrl), term, F64_TOL) <= 1e-9
    # [E][P][3] parameters, as plan_lattice returns them
    assert np.array_equal(vm.rollout_spiral(g["state0"], g["params"].reshape(4, 7, 3), H, torque=tq), term)


def test_spiral_config3_full_size_vs_oracle_and_properties(gpu_vm, oracle, workloads):
    """65536 rollouts (9363 egos x 7 lattice paths) x 200 steps: fp32 vs the fp64 oracle on ALL rollouts;
    determinism; split horizon is NOT a property here (the arc length restarts with the launch);
    host ABI == device ABI; trajectory rows == chained shorter launches' terminals."""
    import torch
    n, H, dt = 65536, 200, 1e-3
    s0, sp = workloads.config3_spiral(n, H, np.float32)
    vm = gpu_vm(dt)
    dev = torch.device("cuda:0")
    s0d, spd = torch.from_numpy(s0).to(dev), torch.from_numpy(sp).to(dev)
    term, traj = vm.rollout_spiral(s0d, spd, H, traj_stride=50)
    torch.cuda.synchronize()
    th = term.cpu().numpy()
    want, wtraj = oracle.rollout_spiral(oracle.default_params(), s0.astype(np.float64), sp.astype(np.float64), H, dt,
                                        traj_stride=50, nthreads=oracle.max_threads())
    e = parity(th, want, F32_TOL, "spiral config3 fp32")
    assert np.abs(th - want).max() <= 1e-3
    parity(traj.cpu().numpy(), wtraj, F32_TOL, "spiral config3 fp32 trajectory")
    print(f"\n  spiral config3: fp32 row-relative err {e:.2e}, max-abs {np.abs(th - want).max():.2e}")
    assert torch.equal(term, vm.rollout_spiral(s0d, spd, H)), "determinism"
    assert torch.equal(traj[-1], term) and torch.equal(traj[1], vm.rollout_spiral(s0d, spd, 100))
    assert np.array_equal(vm.rollout_spiral(s0[:, :4099], sp[:4099], H), th[:, :4099]), "host ABI == device ABI"
    # fp64 on a subsample
    k = 7 * 300
    t64 = vm.rollout_spiral(s0[:, :k].astype(np.float64), sp[:k].astype(np.float64), H)
    assert parity(t64, want[:, :k], F64_TOL, "spiral fp64") <= 1e-9


def test_spiral_steering_limit_and_edges(gpu_vm, oracle, workloads):
    n, H, dt = 7 * 40, 120, 1e-3
    s0, sp = workloads.config3_spiral(n, H, np.float64)
    sp[:, :2] *= 12.0                                              # tight spirals: atan(L kappa) up to ~30 degrees
    p = oracle.default_params()
    vm = gpu_vm(dt)
    for lim in (np.deg2rad(30), np.deg2rad(3), 0.0, np.pi):        # default limit, biting limit, locked wheels, no clip
        want = oracle.rollout_spiral(p, s0, sp, H, dt, max_steer=lim, torque=50.0)
        got = vm.rollout_spiral(s0, sp, H, torque=50.0, max_steer=lim)
        assert parity(got, want, F64_TOL, f"max_steer {lim}") <= 1e-9
        g32 = vm.rollout_spiral(s0.astype(np.float32), sp.astype(np.float32), H, torque=50.0, max_steer=lim)
        parity(g32, want, F32_TOL, f"fp32 max_steer {lim}")
    # H = 0 and empty batches
    assert np.array_equal(vm.rollout_spiral(s0, sp, 0), s0)
    assert vm.rollout_spiral(s0[:, :0], sp[:0], 5).shape == (12, 0)
    with pytest.raises(ValueError):
        vm.rollout_spiral(s0, sp[:-1], H)
    # a lane beyond the FAST range (|yaw| > 2^16 rad) takes the SAFE step, which turns the tangent back
    # into the angle: still the oracle's result, and its neighbours are untouched
    s1 = s0.astype(np.float32).copy()
    s1[7, 5] = 70000.0
    want = oracle.rollout_spiral(p, s1.astype(np.float64), sp, H, dt)
    got = vm.rollout_spiral(s1, sp.astype(np.float32), H)
    ref = vm.rollout_spiral(s0.astype(np.float32), sp.astype(np.float32), H)
    parity(np.delete(got, 7, axis=0), np.delete(want, 7, axis=0), 2e-2, "SAFE lane")      # fp32 yaw = 7e4: 4e-3 rad ulp
    mask = np.ones(n, bool)
    mask[5] = False
    assert np.array_equal(got[:, mask], ref[:, mask])


def test_plan_rollout_select_pipeline_on_device(gpu_vm, oracle):
    """plan_lattice -> rollout_spiral -> select_best_rollout without leaving the GPU: the dynamic rollouts
    of the planned spirals feed the collision check / best-path selection in place; every stage against the
    oracle fed the device's own spiral parameters."""
    import torch
    dev = torch.device("cuda:0")
    E, P, H, dt = 24, 7, 500, 2e-3          # 1 s at 20 m/s: two thirds of the 30 m spirals (dt as BASELINE configs[4])
    th = np.linspace(0.0, 2 * np.pi, 4000, endpoint=False)
    gpx, gpy = 200.0 * np.cos(th), 200.0 * np.sin(th)
    k = np.random.default_rng(12).integers(0, 4000, E)
    ego = np.stack([gpx[k] + 0.3, gpy[k] - 0.3, th[k] + np.pi / 2 + 0.03])
    vm = gpu_vm(dt)
    lat = vm.plan_lattice(torch.from_numpy(gpx).to(dev), torch.from_numpy(gpy).to(dev), torch.from_numpy(ego).to(dev), 25.0)
    U0 = 20.0
    s0 = np.zeros((12, E * P))
    s0[0], s0[3:7] = U0, U0 / oracle.default_params().rw
    s0[8], s0[9], s0[7] = np.repeat(ego[0], P), np.repeat(ego[1], P), np.repeat(ego[2], P)
    s0d = torch.from_numpy(s0).to(dev)
    term, traj = vm.rollout_spiral(s0d, lat["params"], H, torque=0.0, traj_stride=25)
    gi = lat["goal_index"].long().cpu().numpy()
    goal = np.stack([gpx[gi], gpy[gi]])
    obst = np.stack([gpx[k[::3]] * 1.004, gpy[k[::3]] * 1.004], axis=1) + np.array([4.0, 3.0])
    free, best, score = vm.select_best_rollout(traj, P, torch.from_numpy(obst).to(dev), torch.from_numpy(goal).to(dev))
    torch.cuda.synchronize()
    params = lat["params"].cpu().numpy()
    wterm, wtraj = oracle.rollout_spiral(oracle.default_params(), s0, params.reshape(-1, 3), H, dt, torque=0.0, traj_stride=25)
    assert parity(term.cpu().numpy(), wterm, F64_TOL, "pipeline rollout") <= 1e-8
    assert parity(traj.cpu().numpy(), wtraj, F64_TOL, "pipeline trajectory") <= 1e-8
    # the rollouts do follow their spirals: after the spiral's length the vehicle sits near the lattice goal
    assert np.isfinite(score.cpu().numpy()[best.cpu().numpy() >= 0]).all()
    # selection against the oracle on the device's own trajectories (ego-major [L][12][N] consumed in place)
    tr = traj.cpu().numpy()
    L = tr.shape[0]
    paths = np.stack([tr[:, 8], tr[:, 9], tr[:, 7]], axis=0)                    # [3][L][N]
    paths = np.ascontiguousarray(np.transpose(paths.reshape(3, L, E, P), (2, 3, 0, 1)))   # [E][P][3][L]
    ofree, obest, oscore = oracle.select_best_path(paths, obst, goal)
    assert np.array_equal(free.cpu().numpy().astype(bool), ofree.astype(bool))
    assert np.array_equal(best.cpu().numpy(), obest)
    assert (free.cpu().numpy().sum(axis=1) < P).any(), "the obstacles must block some rollouts for the test to mean anything"


def test_g13_selection_with_dropped_spirals(gpu_vm, oracle):
    """The reference drops unreachable spirals before the collision check and the selection
    (local_planner.py:312-321,366-384): vdyn_select_best_path_*'s validity argument, against the reference's
    own planning cycles (G13), incl. 'every surviving path collides' and 'no spiral survives' (best -1), on
    host and device pointers, fp64 and fp32; and interpolate_waypoints keeping the previous table then."""
    import torch
    g = load_golden("g13_dropped_spirals.npz")
    vm = gpu_vm(1e-3)
    n = len(g["names"])
    goal = g["goal"].T.copy()
    keep = [np.flatnonzero(v) for v in g["validity"]]
    want = np.array([-1 if b < 0 else keep[i][b] for i, b in enumerate(g["best_kept"])])
    args = (g["circle_offsets"], g["circle_radii"], float(g["weight"]))
    free, bi, bs = vm.select_best_path(g["paths"], g["obstacles"], goal, *args, validity=g["validity"])
    assert np.array_equal(free, g["free_full"]) and np.array_equal(bi, want)
    assert want.tolist()[2] == -1 and want.tolist()[4] == -1 and np.isinf(bs[[2, 4]]).all()
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    ft, bt, _ = vm.select_best_path(t(g["paths"]), t(g["obstacles"]), t(goal), *args,
                                    validity=t(g["validity"].astype(np.int32)))
    assert np.array_equal(ft.cpu().numpy(), g["free_full"]) and np.array_equal(bt.cpu().numpy(), want)
    f32, b32, _ = vm.select_best_path(g["paths"].astype(np.float32), g["obstacles"].astype(np.float32),
                                      goal.astype(np.float32), *args, validity=g["validity"])
    assert np.array_equal(f32, g["free_full"]) and np.array_equal(b32, want)
    # the oracle agrees case by case (scores too)
    for i in range(n):
        of, ob, os_ = oracle.select_best_path(g["paths"][i][None], g["obstacles"][i], goal[:, i:i + 1], *args,
                                              validity=g["validity"][i][None])
        assert ob[0] == bi[i] and (np.isinf(bs[i]) if ob[0] < 0 else abs(os_[0] - bs[i]) <= 1e-9)
    # an ego without a selectable path keeps the Stanley table of the previous cycle
    wp, wc = vm.interpolate_waypoints(g["paths"], np.array([3, 1, 1, 2, 3], np.int32), 0.01, 4096)
    before = (wp.copy(), wc.copy())
    vm.interpolate_waypoints(g["paths"], bi.astype(np.int32), 0.01, 4096, out=(wp, wc))
    for i in range(n):
        if want[i] < 0:
            assert wc[i] == before[1][i] and np.array_equal(wp[i], before[0][i])
        else:
            one, onec = vm.interpolate_waypoints(g["paths"][i:i + 1], bi[i:i + 1].astype(np.int32), 0.01, 4096)
            assert wc[i] == onec[0] and np.array_equal(wp[i, :wc[i]], one[0, :onec[0]])
