"""Synthetic input definitions for the BASELINE.json configurations.

Pure NumPy, no GPU and no oracle dependency: the golden-vector generator,
the parity tests and bench.py all build their inputs from here so that every
leg (reference, oracle, HIP) sees bit-identical inputs.

Conventions:

* persistent rollout state, SoA ``[12][N]``, rows
  ``U, V, wz, wFL, wFR, wRL, wRR, yaw, x, y, ax_prev, ay_prev``
  (the 10 integrated states of vehicle_model.py:224 plus the two
  accelerations the caller carries step to step, drive.py:141);
* 2-scalar controls ``(delta_front, torque_all)`` expanding to the
  call pattern of drive.py:142-143: ``delta=[d,d,0,0]``,
  ``tire_torques=[t,t,t,t]``, ``mu_max=[1,1,1,1]``;
* per-rollout controls are time-major ``[H][2][N]``; shared controls are a
  table ``[P][H][2]`` plus ``path_id[N]``.
"""
from __future__ import annotations

import numpy as np

NSTATE = 12  # 10 integrated + ax_prev + ay_prev
ROW_NAMES = ("U", "V", "wz", "wFL", "wFR", "wRL", "wRR", "yaw", "x", "y",
             "ax_prev", "ay_prev")

# rw of the default vehicle (vehicle_model.py:38: rr - (mf/2 + mus)/kf)
DEFAULT_RW = 0.329 - (987.89 / 2 + 50) / 26290

NUM_PATHS = 7        # drive.py:21
PATH_OFFSET = 2.0    # drive.py:24


def straight_state(U, n, rw=DEFAULT_RW, dtype=np.float64):
    """``[U,0,0,U/rw x4,0,0,0]`` with zero carried accelerations
    (drive.py:60-65), replicated n times, SoA [12][n]."""
    s = np.zeros((NSTATE, n), dtype=np.float64)
    s[0] = U
    s[3:7] = U / rw
    return s.astype(dtype)


def config2(n_side=64, H=200, dtype=np.float64):
    """Config 2: n_side**2 identical-vehicle rollouts, constant controls from
    a fixed (delta, torque) grid; rollout r -> (delta[r // n_side],
    torque[r % n_side]).  No RNG."""
    n = n_side * n_side
    d = np.linspace(-0.3, 0.3, n_side)
    t = np.linspace(-200.0, 400.0, n_side)
    r = np.arange(n)
    ctrl1 = np.stack([d[r // n_side], t[r % n_side]])          # [2][n]
    ctrl = np.broadcast_to(ctrl1, (H, 2, n)).astype(dtype)     # [H][2][n]
    return straight_state(25.0, n, dtype=dtype), np.ascontiguousarray(ctrl)


def _ego_states(rng, n_ego, rw=DEFAULT_RW):
    """Ego draw order is part of the workload definition: U, V, wz,
    wheel-speed perturbation [4][E], yaw, x, y."""
    U = rng.uniform(10.0, 30.0, n_ego)
    V = rng.normal(0.0, 0.2, n_ego)
    wz = rng.normal(0.0, 0.1, n_ego)
    wp = rng.uniform(-0.01, 0.01, (4, n_ego))
    yaw = rng.uniform(-np.pi, np.pi, n_ego)
    x = rng.uniform(0.0, 100.0, n_ego)
    y = rng.uniform(0.0, 100.0, n_ego)
    s = np.zeros((NSTATE, n_ego))
    s[0], s[1], s[2] = U, V, wz
    s[3:7] = U / rw * (1.0 + wp)
    s[7], s[8], s[9] = yaw, x, y
    return s


def lattice_controls(H=200, num_paths=NUM_PATHS, dtype=np.float64):
    """Shared per-lattice-path control table [P][H][2]:
    delta_k[t] = 0.02 (k-3) sin(2 pi t / H) rad, torque 100 N m."""
    k = np.arange(num_paths)[:, None] - num_paths // 2
    t = np.arange(H)[None, :]
    tab = np.empty((num_paths, H, 2))
    tab[:, :, 0] = 0.02 * k * np.sin(2.0 * np.pi * t / H)
    tab[:, :, 1] = 100.0
    return tab.astype(dtype)


def config3(n=65536, H=200, dtype=np.float32, seed=20240):
    """Config 3: rollout r -> ego r // 7, lattice path r % 7 (the last ego is
    partial when 7 does not divide n).  Returns state0 [12][n], shared control
    table [7][H][2], path_id [n] int32."""
    n_ego = -(-n // NUM_PATHS)
    ego = _ego_states(np.random.default_rng(seed), n_ego)
    r = np.arange(n)
    state0 = ego[:, r // NUM_PATHS]
    path_id = (r % NUM_PATHS).astype(np.int32)
    return (np.ascontiguousarray(state0.astype(dtype)),
            lattice_controls(H, dtype=dtype), path_id)


def config3_spiral(n=65536, H=200, dtype=np.float32, seed=20240, seed_sf=20245):
    """Config 3 with lattice-driven steering ("realistic alternative"): the same
    egos and rollout -> (ego r // 7, path r % 7) map as ``config3``; rollout r follows a cubic spiral
    (p1, p2, sf) that changes lane by (k - 3) * 2 m (drive.py:21,24) over sf ~ U[25, 35] m per ego:
    the antisymmetric spiral p1 = -p2 = q has kappa(s) = 13.5 q u (1 - u)(1 - 2u), u = s / sf, and a
    lateral end offset of 0.225 q sf^2.  Returns state0 [12][n], spiral [n][3].  (H is unused: the
    steering comes from the spiral, not from a table.)"""
    state0, _, path_id = config3(n, H, dtype, seed)
    n_ego = -(-n // NUM_PATHS)
    sf_e = np.random.default_rng(seed_sf).uniform(25.0, 35.0, n_ego)
    r = np.arange(n)
    sf = sf_e[r // NUM_PATHS]
    q = (path_id - NUM_PATHS // 2) * PATH_OFFSET / (0.225 * sf * sf)
    spiral = np.stack([q, -q, sf], axis=1)
    return state0, np.ascontiguousarray(spiral.astype(dtype))


def expand_shared_controls(table, path_id):
    """[P][H][2] + path_id[N] -> per-rollout time-major [H][2][N]."""
    return np.ascontiguousarray(np.transpose(table[path_id], (1, 2, 0)))


def config5(E=1024, C=512, H=50, dtype=np.float32,
            seed_ego=20241, seed_cand=20242):
    """Config 5 (MPC): E egos x C shared control-sequence candidates x H steps.
    Candidates: 5 knots x (H/5)-step hold, delta = clip(N(0,0.05), +-30 deg),
    torque = 100 + N(0,200).  Returns ego [12][E], cand [H][2][C],
    goal [2][E]."""
    rng = np.random.default_rng(seed_ego)
    ego = _ego_states(rng, E)
    lat = rng.uniform(-0.5, 0.5, E)
    knots = 5
    hold = H // knots
    rc = np.random.default_rng(seed_cand)
    dk = np.clip(rc.normal(0.0, 0.05, (knots, C)), -0.5236, 0.5236)
    tk = 100.0 + rc.normal(0.0, 200.0, (knots, C))
    cand = np.empty((H, 2, C))
    cand[:, 0, :] = np.repeat(dk, hold, axis=0)[:H]
    cand[:, 1, :] = np.repeat(tk, hold, axis=0)[:H]
    horizon_t = H * 2e-3
    goal = np.stack([
        ego[8] + ego[0] * horizon_t * np.cos(ego[7]) - lat * np.sin(ego[7]),
        ego[9] + ego[0] * horizon_t * np.sin(ego[7]) + lat * np.cos(ego[7]),
    ])
    return ego.astype(dtype), cand.astype(dtype), goal.astype(dtype)


def closed_loop_config(n=65536, P=NUM_PATHS, W=1024, ds=0.03, dtype=np.float32, seed=20243):
    """Closed-loop tracking workload: P constant-curvature
    waypoint tables of W points at `ds` spacing (kappa_k = 0.004 (k - P//2) 1/m, all starting
    at the origin heading +x), n vehicles spread over the first metres of their table with
    lateral / heading / speed perturbations.  Returns state0 [12][n], cstate0 [6][n]
    (x_del, total_vel_error, prev_vel, target_vel, delta, torque), wp [P][W][2], wcount [P],
    path_id [n]."""
    rng = np.random.default_rng(seed)
    s_arc = np.arange(W) * ds
    wp = np.empty((P, W, 2))
    for k in range(P):
        kap = 0.004 * (k - P // 2)
        if kap == 0.0:
            wp[k, :, 0], wp[k, :, 1] = s_arc, 0.0
        else:
            wp[k, :, 0] = np.sin(kap * s_arc) / kap
            wp[k, :, 1] = (1.0 - np.cos(kap * s_arc)) / kap
    pid = (np.arange(n) % P).astype(np.int32)
    kap = 0.004 * (pid - P // 2)
    s_along = rng.uniform(0.0, 5.0, n)
    lat = rng.normal(0.0, 0.3, n)
    th = kap * s_along
    cx = np.where(kap == 0.0, s_along, np.sin(th) / np.where(kap == 0.0, 1.0, kap))
    cy = np.where(kap == 0.0, 0.0, (1.0 - np.cos(th)) / np.where(kap == 0.0, 1.0, kap))
    U = rng.uniform(15.0, 30.0, n)
    st = np.zeros((NSTATE, n))
    st[0] = U
    st[3:7] = U / DEFAULT_RW
    st[7] = th + rng.normal(0.0, 0.02, n)
    st[8] = cx - lat * np.sin(th)
    st[9] = cy + lat * np.cos(th)
    cs = np.zeros((6, n))
    cs[2] = U          # prev_vel (drive.py:50)
    cs[3] = 25.0       # target_vel (drive.py:46,51)
    return (st.astype(dtype), cs.astype(dtype), wp.astype(dtype), np.full(P, W, dtype=np.int32), pid)


MPC_W_DELTA = 1e-3  # weight of sum(delta^2) in the config-5 cost


def shard_egos(n_units, world_size, rank, group=NUM_PATHS):
    """Contiguous blocks of whole egos per rank: unit
    range [lo, hi) of rank `rank`; every rank but the last gets
    ceil(E / world) egos of `group` units."""
    n_ego = -(-n_units // group)
    per = -(-n_ego // world_size)
    lo = min(rank * per * group, n_units)
    hi = min((rank + 1) * per * group, n_units)
    return lo, hi
