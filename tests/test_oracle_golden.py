"""CPU: pin the oracle (oracle/) against golden vectors produced by importing
the reference (tests/golden/generate_golden.py).  The oracle restates
vehicle_model.py:220-445 operation for operation, so fp64 agreement is at
rounding level (libm vs NumPy's sin/cos/atan kernels differ by <= 1 ulp)."""
import numpy as np
import pytest

from conftest import load_golden

TOL = 1e-12  # relative, vs the magnitude of each quantity's row


def close(a, b, tol=TOL, scale=None):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    s = np.maximum(np.abs(b), 1.0 if scale is None else scale)
    err = np.max(np.abs(a - b) / s)
    assert err <= tol, err


def test_kat1_bit_exact(oracle):
    """KAT-1 of SURVEY.md section 8a, measured on the reference."""
    p = oracle.default_params()
    s = [25, 0, 0] + [25 / p.rw] * 4 + [0, 0, 0]
    o = oracle.planar_model_RK4(p, 1e-4, s, [0] * 4, [1] * 4, [0.02, 0.02, 0, 0], 0, 0)
    want = [2.4999996979914723e+01, 2.9402126600312881e-04, 3.7460500999973191e-04,
            8.1086421344188210e+01, 8.1086469091352370e+01, 8.1087243497252814e+01,
            8.1087288891381220e+01, 1.8738250992909321e-08, 2.4999998502186231e-03,
            1.4726234923492799e-08]
    close(o[0], want, 1e-15)
    assert abs(o[7] - -0.030200889513079934) < 1e-15
    assert abs(o[8] - 2.944897222404597) < 1e-14
    assert abs(p.rw - 0.308309813617345) < 1e-15
    assert abs(p.Izz - 1948.2304506781593) < 1e-9


def test_g1_step_kats(oracle):
    g = load_golden("g1_step_kat.npz")
    p = oracle.default_params()
    for i in range(len(g["state"])):
        o = oracle.planar_model_RK4(p, float(g["dt"]), g["state"][i], g["torque"][i], g["mu"][i],
                                    g["delta"][i], *g["ax_ay_prev"][i])
        close(o[0], g["state_update"][i])
        close([o[1], o[2], o[3], o[4]], g["xyyawU"][i])
        close(o[5], g["state_dot"][i], scale=np.abs(g["state_dot"][i]).max())
        close(o[6], g["outputs"][i], scale=np.abs(g["outputs"][i]).max())
        close([o[7], o[8]], g["acc"][i])


def test_g2_derivative(oracle):
    g = load_golden("g2_deriv.npz")
    p = oracle.default_params()
    for i in range(len(g["state"])):
        o = oracle.planar_model(p, g["state"][i], g["torque"][i], g["mu"][i], g["delta"][i],
                                *g["ax_ay_prev"][i])
        close(o[0], g["state_dot"][i], scale=np.abs(g["state_dot"][i]).max())
        close(o[1:5], g["aux"][i])
        close(o[5], g["outputs"][i], scale=np.abs(g["outputs"][i]).max())
        close(o[6:8], g["acc"][i])


@pytest.mark.parametrize("tag,dt", [("dt1e-4", 1e-4), ("dt1e-3", 1e-3)])
def test_g3_rollouts_cfg2(oracle, tag, dt):
    g = load_golden("g3_rollout_cfg2.npz")
    p = oracle.default_params()
    term, traj = oracle.rollout(p, g["state0"], g["ctrl"], dt, traj_stride=20, nthreads=4)
    close(term, g["terminal_" + tag], 1e-10)
    close(traj, g["every20_" + tag], 1e-10)


@pytest.mark.parametrize("tag", ["world", "waypoints"])
def test_g4_closed_loop_replay(oracle, tag):
    """Every RK4 call the reference's Car.drive made in 3 frames (drive.py:141-143),
    replayed call by call with the logged inputs."""
    g = load_golden(f"g4_closed_loop_{tag}.npz")
    p = oracle.default_params()
    dt = float(g["dt"])
    for i in range(len(g["state"])):
        o = oracle.planar_model_RK4(p, dt, g["state"][i], g["torque"][i], g["mu"][i],
                                    g["delta"][i], *g["ax_ay_prev"][i])
        close(o[0], g["state_update"][i])
        close(o[5], g["state_dot"][i], scale=np.abs(g["state_dot"][i]).max())
        close(o[6], g["outputs"][i], scale=np.abs(g["outputs"][i]).max())
        close([o[7], o[8]], g["acc"][i])
    # and as one uninterrupted rollout with the logged zero-order-hold controls (k = 12)
    n = len(g["state"])
    ctrl = np.concatenate([g["delta"], g["torque"], g["mu"]], axis=1)[:, :, None]  # [H][12][1]
    s0 = np.concatenate([g["state"][0], g["ax_ay_prev"][0]])[:, None]
    term = oracle.rollout(p, s0, ctrl, dt)
    close(term[:10, 0], g["state_update"][n - 1], 1e-11)
    close(term[10:, 0], g["acc"][n - 1], 1e-9)


def test_g5_quirks(oracle):
    g = load_golden("g5_quirks.npz")
    p = oracle.default_params()
    H, dt = int(g["H"]), float(g["dt"])
    for i, name in enumerate(g["names"]):
        c12 = np.concatenate([g["delta"][i], g["torque"][i], g["mu"][i]])
        ctrl = np.broadcast_to(c12[None, :, None], (H, 12, 1))
        term = oracle.rollout(p, g["state0"][i][:, None], ctrl, dt)[:, 0]
        o = oracle.planar_model_RK4(p, dt, g["state0"][i][:10], g["torque"][i], g["mu"][i],
                                    g["delta"][i], g["state0"][i][10], g["state0"][i][11])
        first = np.concatenate([o[0], [o[7], o[8]], o[5], o[6]])
        f_ref = g["first_step"][i]
        close(first[:12], f_ref[:12], 1e-12), name
        close(first[12:22], f_ref[12:22], 1e-12, scale=np.abs(f_ref[12:22]).max())
        close(first[22:], f_ref[22:], 1e-12, scale=np.abs(f_ref[22:]).max())
        close(term, g["terminal"][i], 1e-10)


def test_g5_zero_slip_takes_fallback_branch(oracle):
    """Quirk Q5: combined slip exactly 0 -> vehicle_model.py:311-348 fallback."""
    g = load_golden("g5_quirks.npz")
    i = list(g["names"]).index("Q5_zero_slip")
    p = oracle.default_params()
    o = oracle.planar_model(p, g["state0"][i][:10], g["torque"][i], g["mu"][i], g["delta"][i], 0, 0)
    assert o[5][12] == 0.0 and o[5][15] == 0.0          # s_FL, s_RR
    assert np.all(np.isfinite(o[0])) and np.all(o[5][:8] == 0.0)


def test_g6_float32_floor(oracle):
    """Informational fp32 floor: the reference fed float32 arrays vs the float
    instantiation of the oracle (libm sinf/atanf vs NumPy's float32 kernels
    differ in the last ulp, which the 200-step recurrence amplifies)."""
    g3 = load_golden("g3_rollout_cfg2.npz")
    g6 = load_golden("g6_ref_float32.npz")
    p = oracle.default_params()
    for tag, dt in (("dt1e-4", 1e-4), ("dt1e-3", 1e-3)):
        term = oracle.rollout(p, g3["state0"].astype(np.float32), g3["ctrl"].astype(np.float32), dt)
        assert term.dtype == np.float32
        close(term, g6["terminal_" + tag], 1e-3)
        close(term, g3["terminal_" + tag], 1e-3)         # and vs the fp64 reference


def test_g7_mpc(oracle):
    g = load_golden("g7_mpc.npz")
    p = oracle.default_params()
    ego, cand, goal = (g[k].astype(np.float64) for k in ("ego", "cand", "goal"))
    bc, bi, cost = oracle.mpc_argmin(p, ego, cand, goal, float(g["dt"]), float(g["w_delta"]),
                                     return_costs=True)
    close(cost, g["cost"], 1e-10)
    assert np.array_equal(bi, g["best_idx"])
    close(bc, g["best_cost"], 1e-10)
    # terminal states of all E*C rollouts through the generic rollout entry
    E, C = ego.shape[1], cand.shape[2]
    s0 = np.repeat(ego, C, axis=1)
    ctrl = np.tile(cand, (1, 1, E))
    term = oracle.rollout(p, s0, ctrl, float(g["dt"]))
    close(term.reshape(12, E, C), g["terminal"], 1e-10)


def test_g8_rollouts_cfg3_shared_controls(oracle, workloads):
    g = load_golden("g8_rollout_cfg3.npz")
    p = oracle.default_params()
    s0, tab = g["state0"].astype(np.float64), g["table"].astype(np.float64)
    term, traj = oracle.rollout(p, s0, tab, float(g["dt"]), path_id=g["path_id"], traj_stride=50)
    close(term, g["terminal"], 1e-10)
    close(traj, g["every50"], 1e-10)
    # shared table == expanded per-rollout controls, bit for bit
    term2 = oracle.rollout(p, s0, workloads.expand_shared_controls(tab, g["path_id"]), float(g["dt"]))
    assert np.array_equal(term, term2)


def test_vectorised_numpy_restatement_against_reference_vectors():
    """oracle/numpy_batch.py (the "what a NumPy user would write" CPU line of bench.py) against
    the reference's own outputs: G2 derivatives (random states, k = 12 inputs) and the G8
    lattice rollouts (terminal states after 200 steps)."""
    from oracle import numpy_batch as NB
    p = NB.Params()
    g = load_golden("g2_deriv.npz")
    sd, axc, ayc = NB.planar_model(p, g["state"].T.copy(), g["torque"].T.copy(), g["mu"].T.copy(),
                                   g["delta"].T.copy(), g["ax_ay_prev"][:, 0], g["ax_ay_prev"][:, 1])
    scale = np.abs(g["state_dot"]).max(axis=1)
    assert (np.abs(sd.T - g["state_dot"]).max(axis=1) <= 1e-11 * scale).all()
    assert np.abs(np.stack([axc, ayc], axis=1) - g["acc"]).max() <= 1e-10
    g = load_golden("g8_rollout_cfg3.npz")
    term = NB.rollout(p, g["state0"].astype(np.float64), g["table"].astype(np.float64), float(g["dt"]), g["path_id"])
    close(term, g["terminal"], 1e-9)


def test_threads_do_not_change_results(oracle, workloads):
    p = oracle.default_params()
    s0, ctrl = workloads.config2(8, 30)
    a = oracle.rollout(p, s0, ctrl, 1e-3, nthreads=1)
    b = oracle.rollout(p, s0, ctrl, 1e-3, nthreads=4)
    assert np.array_equal(a, b)


def test_empty_and_bad_inputs(oracle):
    p = oracle.default_params()
    term = oracle.rollout(p, np.zeros((12, 0)), np.zeros((5, 2, 0)), 1e-3)
    assert term.shape == (12, 0)
    with pytest.raises(ValueError):
        oracle.rollout(p, np.zeros((10, 3)), np.zeros((5, 2, 3)), 1e-3)
    with pytest.raises(ValueError):
        oracle.rollout(p, np.zeros((12, 3)), np.zeros((5, 3, 3)), 1e-3)
    with pytest.raises(ValueError):
        oracle.rollout(p, np.zeros((12, 3)), np.zeros((2, 5, 2)), 1e-3, path_id=[0, 1, 2])


# ---- "next" row 1: controllers (G9) -------------------------------------------------------
def test_g9_stanley_and_pid_calls(oracle):
    """Every stanley_control / long_control call of 3 frames of the reference's Car.drive."""
    g = load_golden("g9_closed_loop_controls.npz")
    gains = g["gains"]
    cp = oracle.ctrl_params(k=gains[0], k_soft=gains[1], max_steer=gains[2], lookahead=gains[3],
                            deadband=gains[4], kp=gains[5], ki=gains[6], kd=gains[7])
    for i in range(len(g["stanley_in"])):
        f = int(g["stanley_wp"][i])
        wp = g["waypoints"][f, :g["waypoint_count"][f]]
        d, idx, cte = oracle.stanley_control(cp, wp, *g["stanley_in"][i])
        assert idx == int(g["stanley_out"][i, 1])
        assert abs(d - g["stanley_out"][i, 0]) <= 1e-12 and abs(cte - g["stanley_out"][i, 2]) <= 1e-12
        tot, tau = oracle.long_control(cp, *g["pid_in"][i])
        assert abs(tot - g["pid_out"][i, 0]) <= 1e-15 and abs(tau - g["pid_out"][i, 1]) <= 1e-11


def test_g9_closed_loop_three_frames(oracle):
    """The whole closed loop (controllers + filter + RK4) frame by frame with the planner's
    waypoint lists as inputs reproduces the reference's 300-step trajectory (G4/G9)."""
    g = load_golden("g9_closed_loop_controls.npz")
    gains = g["gains"]
    cp = oracle.ctrl_params(k=gains[0], k_soft=gains[1], max_steer=gains[2], lookahead=gains[3],
                            deadband=gains[4], kp=gains[5], ki=gains[6], kd=gains[7])
    p = oracle.default_params()
    dt = float(g["dt"])
    s = np.concatenate([g["state"], g["ax_ay_prev"]])[:, None]
    c = np.array([g["x_del"], g["total_vel_error"], g["prev_vel"], g["target_vel"], 0.0, 0.0])[:, None]
    for f in range(3):
        wp = g["waypoints"][f:f + 1, :, :2].copy()
        wp[np.isnan(wp)] = 0.0
        s, c, log = oracle.closed_loop(p, cp, s, c, wp, g["waypoint_count"][f:f + 1], [0], dt, 100, log=True)
        want = g["rk4_log"][f * 100:(f + 1) * 100]                      # state12, delta, torque
        close(log[:, :12, 0], want[:, :12], 1e-10)
        close(log[:, 12, 0], want[:, 12], 1e-11)
        close(log[:, 13, 0], want[:, 13], 1e-9)
        idx = g["stanley_out"][f * 10:(f + 1) * 10, 1]
        assert np.array_equal(log[::10, 14, 0], idx)
    assert abs(c[0, 0] - g["x_del_log"][-1]) <= 1e-14


# ---- "next" row 2: collision check + best-path selection (G10) --------------------------------
def test_g10_collision_and_selection(oracle):
    g = load_golden("g10_collision_select.npz")
    E = len(g["paths"])
    free, bi, bs = oracle.select_best_path(g["paths"], g["obstacles"], g["goal"].T.copy(),
                                           g["circle_offsets"], g["circle_radii"], float(g["weight"]))
    assert np.array_equal(free, g["collision_free"])
    assert np.array_equal(bi, g["best_index"])
    assert np.isinf(bs[bi < 0]).all() and np.isfinite(bs[bi >= 0]).all()
    # shared obstacle set == per-ego copies of it
    f2, b2, s2 = oracle.select_best_path(g["paths"][:3], g["obstacles"][0], g["goal"][:3].T.copy())
    assert np.array_equal(f2, g["collision_free"][:3]) and np.array_equal(b2, g["best_index"][:3])
    assert E == 63


# ---- "next" row 3: lattice generation (G11) --------------------------------------------------------
def test_g11_planning_cycles(oracle):
    """3 planning cycles of the reference's Car.drive: indices exact; goal states, sampled
    spirals and transformed paths at rounding level when fed the reference's own optimum;
    the restated objective reproduces L-BFGS-B's optimum through the same SciPy call."""
    g = load_golden("g11_lattice.npz")
    look, P, off, res = g["consts"]
    for f in range(3):
        o = oracle.plan_paths(g["px"], g["py"], g["ego"][f], float(g["target_vel"]), look, int(P), off,
                              spiral_params=g["opt_x"][f])
        assert o["closest_index"] == g["closest_index"][f] and o["goal_index"] == g["goal_index"][f]
        assert abs(o["closest_len"] - g["closest_len"][f]) <= 1e-14
        close(o["goal_set"], g["goal_set"][f], 1e-12)
        close(o["spiral_x"], g["spiral_x"][f], 1e-12)
        close(o["spiral_y"], g["spiral_y"][f], 1e-12)
        close(o["spiral_t"], g["spiral_t"][f], 1e-12)
        assert np.array_equal(o["validity"], g["validity"][f])
        close(o["paths"], g["transformed"][f], 1e-12)
        full = oracle.plan_paths(g["px"], g["py"], g["ego"][f], float(g["target_vel"]), look, int(P), off)
        close(full["params"], g["opt_x"][f], 1e-6)
        close(full["paths"], g["transformed"][f], 1e-6)


def test_g11_objective_optimum_and_samples(oracle):
    g = load_golden("g11_lattice.npz")
    for i in range(len(g["direct_goals"])):
        xf, yf, tf = g["direct_goals"][i]
        J, grad = oracle.spiral_objective(g["direct_x"][i], xf, yf, tf)
        # finite-difference check of the restated gradient
        for k in range(3):
            h = 1e-6 * max(1.0, abs(g["direct_x"][i][k]))
            pp, pm = g["direct_x"][i].copy(), g["direct_x"][i].copy()
            pp[k] += h
            pm[k] -= h
            fd = (oracle.spiral_objective(pp, xf, yf, tf)[0] - oracle.spiral_objective(pm, xf, yf, tf)[0]) / (2 * h)
            assert abs(fd - grad[k]) <= 1e-5 * max(1.0, abs(grad[k]))
        close(oracle.optimize_spiral(xf, yf, tf), g["direct_x"][i], 1e-6)
        x, y, t = oracle.sample_spiral(g["direct_x"][i])
        close(x, g["direct_sx"][i], 1e-12)
        close(y, g["direct_sy"][i], 1e-12)
        close(t, g["direct_st"][i], 1e-12)


def test_g11_waypoint_interpolation(oracle):
    """The 1 cm re-interpolation the planner hands to the Stanley controller (G9's tables) from
    the best path of each cycle (the index select_best_path_index returned, G10)."""
    g = load_golden("g11_lattice.npz")
    g9 = load_golden("g9_closed_loop_controls.npz")
    g10 = load_golden("g10_collision_select.npz")
    for f in range(3):
        best = g["transformed"][f][g10["best_index"][f]]
        wp = oracle.interpolate_waypoints(best[0], best[1], float(g["target_vel"]), float(g["consts"][3]))
        want = g9["waypoints"][f, :g9["waypoint_count"][f]]
        assert wp.shape == want.shape
        close(wp, want, 1e-12)


def test_g12_spiral_rollouts_oracle_vs_reference(oracle):
    """Lattice-driven rollouts: spiral parameters from the reference's PathOptimizer (L-BFGS-B), the
    steering sequence they imply and the reference's planar_model_RK4 terminal states (G12)."""
    from conftest import rel_err
    g = load_golden("g12_spiral_rollouts.npz")
    p = oracle.default_params()
    H, dt, tq = g["delta"].shape[0], float(g["dt"]), float(g["torque"])
    term, traj, dl = oracle.rollout_spiral(p, g["state0"], g["params"], H, dt, wheelbase=float(g["wheelbase"]),
                                           max_steer=float(g["max_steer"]), torque=tq, traj_stride=20,
                                           return_delta=True)
    assert g["state0"].shape[1] == 28
    assert np.abs(dl - g["delta"]).max() <= 1e-15, "steering sequence of the spiral"
    assert rel_err(term, g["terminal"], 1e-6).max() <= 1e-10
    assert rel_err(traj, g["every20"], 1e-6).max() <= 1e-10
    # the same rollouts through the table-driven oracle path fed the golden delta sequence
    ctrl = np.ascontiguousarray(np.stack([g["delta"], np.full_like(g["delta"], tq)], axis=1))   # [H][2][N]
    assert rel_err(oracle.rollout(p, g["state0"], ctrl, dt), term, 1e-6).max() <= 1e-12
    # clip (stanley_controller.py:128): a 2 degree limit bites on the outer lattice paths
    lim = np.deg2rad(2.0)
    _, d2 = oracle.rollout_spiral(p, g["state0"], g["params"], H, dt, max_steer=lim, torque=tq, return_delta=True)
    assert np.abs(d2).max() <= lim and (np.abs(d2) == lim).any()
    assert np.abs(d2 - np.clip(g["delta"], -lim, lim)).max() <= 1e-15


def _g13_expected_full_index(validity, best_kept):
    """The reference's best_index counts the surviving paths only (local_planner.py:312-321,378)."""
    keep = np.flatnonzero(validity)
    return -1 if best_kept < 0 else int(keep[best_kept])


def test_g13_selection_with_dropped_spirals_oracle_vs_reference(oracle):
    g = load_golden("g13_dropped_spirals.npz")
    assert list(g["names"]) == ["all_valid", "two_dropped", "two_dropped_all_blocked", "two_dropped_free", "none_valid"]
    assert g["validity"].sum(axis=1).tolist() == [7, 5, 5, 5, 0]
    for i in range(len(g["names"])):
        free, bi, _ = oracle.select_best_path(g["paths"][i][None], g["obstacles"][i], g["goal"][i][:, None],
                                              g["circle_offsets"], g["circle_radii"], float(g["weight"]),
                                              validity=g["validity"][i][None])
        assert np.array_equal(free[0], g["free_full"][i]), g["names"][i]
        assert bi[0] == _g13_expected_full_index(g["validity"][i], int(g["best_kept"][i])), g["names"][i]
    # ignoring validity is NOT the reference's result: a dropped spiral would be selectable / would add a penalty
    i = 3                                                                       # two_dropped_free: nothing collides
    _, bi_all, _ = oracle.select_best_path(g["paths"][i][None], g["obstacles"][i], g["goal"][i][:, None],
                                           g["circle_offsets"], g["circle_radii"], float(g["weight"]))
    free_v, _, _ = oracle.select_best_path(g["paths"][i][None], g["obstacles"][i], g["goal"][i][:, None],
                                           g["circle_offsets"], g["circle_radii"], float(g["weight"]),
                                           validity=g["validity"][i][None])
    assert free_v[0].sum() == 5 and bi_all[0] >= 0
