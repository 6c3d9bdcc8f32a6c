"""GPU (-m gpu): the fp32 controllers' own arithmetic (bounded-cost trigonometry, lookahead search on the
cumulative arc length) against the fp64 oracle, call by call.  fp64 stays bit-for-index exact and is
covered in test_gpu_parity.py; fp32 is held to north_star's 1e-3 on states, and its target index may sit
one waypoint off exactly where fp32 rounding decides (stanley_controller.py:56-129)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _tables(rng, P, W, uniform):
    wp = np.zeros((P, W, 2))
    for p in range(P):
        ds = np.full(W, 0.03) if uniform else rng.uniform(0.004, 0.08, W)     # ragged spacing: the guess misses, bisection decides
        s = np.concatenate([[0.0], np.cumsum(ds[1:])])
        kap = 0.01 * (p - P // 2)
        th = kap * s
        wp[p, :, 0] = np.cumsum(np.concatenate([[0.0], ds[1:] * np.cos(th[1:])]))
        wp[p, :, 1] = np.cumsum(np.concatenate([[0.0], ds[1:] * np.sin(th[1:])]))
    return wp


@pytest.mark.parametrize("uniform", [True, False])
def test_fp32_controller_update_vs_fp64_oracle(gpu_vm, pkg, oracle, uniform):
    rng = np.random.default_rng(32 + uniform)
    P, W, n = 5, 900, 4096
    wp = _tables(rng, P, W, uniform)
    wc = np.array([900, 850, 700, 900, 333], dtype=np.int32)
    pid = rng.integers(0, P, n).astype(np.int32)
    # vehicles anywhere along their table, including the last metres (lookahead beyond the path's end)
    k = (rng.uniform(0, 1, n) * (wc[pid] - 1)).astype(int)
    k[: n // 8] = wc[pid[: n // 8]] - 1 - rng.integers(0, 40, n // 8)
    s = np.zeros((12, n))
    s[0] = rng.uniform(5, 30, n)
    s[8] = wp[pid, k, 0] + rng.normal(0, 0.4, n)
    s[9] = wp[pid, k, 1] + rng.normal(0, 0.4, n)
    s[7] = 0.01 * (pid - P // 2) * k * 0.03 + rng.normal(0, 0.2, n) + 2 * np.pi * rng.integers(-3, 4, n)   # unwrapped yaw
    c = np.zeros((6, n))
    c[2], c[3] = s[0], 25.0
    vm = gpu_vm(1e-3)
    cs64, out64 = vm.controller_update(s, c, wp, wcount=wc, path_id=pid)
    cp = oracle.ctrl_params()
    for i in range(0, n, 97):                                   # fp64 device == oracle (exact indices): a sample is enough here
        d, idx, cte = oracle.stanley_control(cp, wp[pid[i], :wc[pid[i]]], s[8, i], s[9, i], s[7, i], s[0, i])
        assert idx == out64[1, i] and abs(d - out64[0, i]) <= 1e-9 and abs(cte - out64[2, i]) <= 1e-9
    cs32, out32 = vm.controller_update(s.astype(np.float32), c.astype(np.float32), wp.astype(np.float32), wcount=wc,
                                       path_id=pid)
    didx = np.abs(out32[1].astype(np.int64) - out64[1].astype(np.int64))
    assert didx.max() <= 1, "fp32 target index at most one waypoint from the fp64 one"
    frac = (didx != 0).mean()
    assert frac <= 0.02, f"{frac:.3%} of fp32 target indices differ"
    same = didx == 0
    # steering angle (clipped to +-30 deg), crosstrack error, filter state, torque
    assert np.abs(out32[0] - out64[0])[same].max() <= 2e-4
    assert np.abs(out32[0] - out64[0]).max() <= 5e-3            # a one-waypoint shift moves the heading error a little
    assert np.abs(out32[2] - out64[2])[same].max() <= 1e-4 * max(1.0, np.abs(out64[2]).max())
    assert np.abs(cs32[0] - cs64[0]).max() <= 1e-5 and np.abs(cs32[5] - cs64[5]).max() <= 1e-2
    print(f"\n  fp32 controllers ({'uniform' if uniform else 'ragged'} tables): {frac:.3%} indices off by one, "
          f"steer err {np.abs(out32[0] - out64[0])[same].max():.1e}")


def test_fp32_closed_loop_indices_follow_fp64(gpu_vm, pkg, oracle, workloads):
    """The bench's closed-loop workload, 4096 vehicles x 200 sub-steps: fp32 log vs fp64 oracle log."""
    n, H, dt = 4096, 200, 1e-3
    st, cs, wp, wc, pid = workloads.closed_loop_config(n, dtype=np.float64)
    ot, oc, olog = oracle.closed_loop(oracle.default_params(), oracle.ctrl_params(), st, cs, wp, wc, pid, dt, H, log=True,
                                      nthreads=oracle.max_threads())
    vm = gpu_vm(dt)
    t32, c32, log = vm.closed_loop(st.astype(np.float32), cs.astype(np.float32), wp.astype(np.float32), H, wcount=wc,
                                   path_id=pid, log=True)
    from conftest import parity
    parity(t32, ot, 1e-3, "closed loop fp32 terminal")
    didx = np.abs(log[::10, 14] - olog[::10, 14])
    assert didx.max() <= 1 and (didx != 0).mean() <= 0.02
    assert np.abs(log[:, 12] - olog[:, 12]).max() <= 2e-3        # filtered steering command


# (seed, controller period in sub-steps, horizon): what tools/closed_loop_seed_sweep.py ran as a one-off in round 3
SWEEP_CASES = ((1, 10, 200), (2, 10, 200), (3, 7, 203), (4, 1, 60), (5, 25, 200), (6, 10, 400))
SWEEP_N = 8192
SWEEP_TABLE = []            # one formatted line per case; test_fp32_closed_loop_seed_sweep_pooled writes them out


def _row_rel(got, want):
    """Per vehicle: the largest |got - want| over the 12 state rows, each relative to its row's scale (conftest.parity's
    measure, kept per vehicle)."""
    w = np.asarray(want, np.float64)
    return (np.abs(np.asarray(got, np.float64) - w) / np.abs(w).max(axis=1, keepdims=True)).max(axis=0)


SWEEP_POOL = {}


def _sweep_case(gpu_vm, oracle, workloads, seed, every, H):
    """One case of the sweep (test_fp32_closed_loop_seed_sweep's docstring says what is asserted); its counts go into
    SWEEP_POOL and its line into SWEEP_TABLE.  Run once per session: the pooled test calls it for whatever case the
    session has not run yet, so it never depends on which tests were selected."""
    if (seed, every, H) in SWEEP_POOL:
        return
    n, dt = SWEEP_N, 1e-3
    st, cs, wp, wc, pid = workloads.closed_loop_config(n, dtype=np.float64, seed=seed)
    f32 = [a.astype(np.float32) for a in (st, cs, wp)]
    P, G = oracle.default_params(), oracle.ctrl_params()
    nt = oracle.max_threads()
    ot, oc, olog = oracle.closed_loop(P, G, st, cs, wp, wc, pid, dt, H, log=True, ctrl_every=every, nthreads=nt)
    of, ocf, oflog = oracle.closed_loop(P, G, *f32, wc, pid, dt, H, log=True, ctrl_every=every, nthreads=nt)
    vm = gpu_vm(dt)
    t64, c64, log64 = vm.closed_loop(st, cs, wp, H, wcount=wc, path_id=pid, log=True, ctrl_every=every)
    t32, c32, log32 = vm.closed_loop(*f32, H, wcount=wc, path_id=pid, log=True, ctrl_every=every)
    assert np.isfinite(t32).all() and np.isfinite(of).all() and np.isfinite(ot).all()

    upd = slice(0, H, every)                                            # the sub-steps with a controller update
    idx64, idx32, idxf, idxo = (l[upd, 14].astype(np.int64) for l in (log64, log32, oflog, olog))
    assert np.array_equal(idx64, idxo), "fp64 target indices must be the oracle's"
    r64 = _row_rel(t64, ot).max()
    assert r64 <= 1e-9, f"fp64 terminal {r64:.2e}"

    rel_k, rel_f = _row_rel(t32, ot), _row_rel(of, ot)                  # kernel fp32 / float oracle, per vehicle
    same_k, same_f = (idx32 == idxo).all(axis=0), (idxf == idxo).all(axis=0)
    out_k, out_f = rel_k > 1e-3, rel_f > 1e-3
    dk = np.abs(idx32 - idxo)
    df = np.abs(idxf - idxo)
    line = (f"seed {seed} every {every:2d} H {H:3d} | fp64: indices exact, terminal {r64:.1e} | "
            f"fp32 kernel: idx differs on {(dk != 0).mean():.4%} of updates (max {int(dk.max())}), "
            f"{int((~same_k).sum()):3d} vehicles with a differing index, {int(out_k.sum()):2d} beyond 1e-3 "
            f"(worst {rel_k.max():.1e}; same-index vehicles worst {rel_k[same_k].max():.1e}) | "
            f"float oracle: idx differs on {(df != 0).mean():.4%} (max {int(df.max())}), "
            f"{int((~same_f).sum()):3d} vehicles, {int(out_f.sum()):2d} beyond 1e-3 "
            f"(worst {rel_f.max():.1e}; same-index worst {rel_f[same_f].max():.1e}) | "
            f"outliers in both: {int((out_k & out_f).sum())}")
    print("\n  " + line, flush=True)
    SWEEP_TABLE.append(line)
    SWEEP_POOL[(seed, every, H)] = (int(out_k.sum()), int(out_f.sum()), int((~same_k).sum()), int((~same_f).sum()), n)

    assert rel_k[same_k].max() <= 1e-3, "a vehicle that steered at the oracle's waypoints all along is off by more than 1e-3"
    assert not (out_k & same_k).any(), "every fp32 outlier must have at least one differing target index"
    assert dk.max() <= 1 and (dk != 0).mean() <= 2 * max((df != 0).mean(), 1e-4)
    assert out_k.sum() <= 2 * out_f.sum() + 3, (int(out_k.sum()), int(out_f.sum()))
    assert np.abs(log32[:, 12] - olog[:, 12])[:, same_k].max() <= 2e-3    # filtered steering command, same-index vehicles


@pytest.mark.parametrize("seed,every,H", SWEEP_CASES)
def test_fp32_closed_loop_seed_sweep(gpu_vm, oracle, workloads, seed, every, H):
    """The fp32 closed loop against the fp64 oracle on six more seeds, update periods 1 / 7 / 10 / 25 and horizons up to
    400 (stanley_controller.py:56-129 every `every` sub-steps, drive.py:128-138), 8192 vehicles each -- and beside it
    the plain-C FLOAT oracle on the same inputs, which is what makes "the outliers are the precision's, not the
    kernel's" checkable:
      * fp64 kernel: target indices exactly the oracle's, terminal states within 1e-9;
      * fp32 kernel: every vehicle whose target indices ALL equal the fp64 oracle's ends within north_star's 1e-3
        (row-relative) of the fp64 oracle;
      * every fp32 vehicle beyond 1e-3 has at least one controller update with another target index than the oracle's
        (the law is discontinuous in the index: that, not the step arithmetic, is what moved it);
      * the share of fp32-kernel vehicles beyond 1e-3 is at most twice the share the float oracle itself puts beyond
        1e-3 of the fp64 oracle (+ 3 vehicles of counting noise per case; the pooled test below has no such slack)."""
    _sweep_case(gpu_vm, oracle, workloads, seed, every, H)


def test_fp32_closed_loop_seed_sweep_pooled(gpu_vm, oracle, workloads):
    """Over the six cases together (49152 vehicles): the kernel's outlier count against the float oracle's, factor 2, no
    additive slack; the table goes to gpurun_out/ (committed copy: profiles/r05_closed_loop_seed_sweep.txt).  Cases the
    session has not run (a `-k` selection) are computed here: this test never passes by skipping."""
    import os
    for case in SWEEP_CASES:
        _sweep_case(gpu_vm, oracle, workloads, *case)
    assert len(SWEEP_POOL) == len(SWEEP_CASES)
    k, f, sk, sf, n = (sum(v[i] for v in SWEEP_POOL.values()) for i in range(5))
    tail = (f"pooled over {n} vehicles: fp32 kernel {k} beyond 1e-3 ({sk} with a differing index), "
            f"plain-C float oracle {f} beyond 1e-3 ({sf} with a differing index); bar: kernel <= 2 x float oracle")
    print("\n  " + tail)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "closed_loop_seed_sweep.txt"), "w") as fh:
            fh.write("# tests/test_gpu_controllers32.py::test_fp32_closed_loop_seed_sweep -- fp32 closed loop vs the fp64 oracle,\n"
                     "# beside the plain-C float oracle on the same inputs (8192 vehicles per case, dt 1e-3)\n")
            fh.write("\n".join(SWEEP_TABLE) + "\n" + tail + "\n")
    assert k <= 2 * f, tail
    assert sk <= 2 * sf, tail


def _adversarial_tables():
    """Six waypoint tables (Wmax = 203: odd, so the LDS image is staged pair by pair) built to break a pruned
    nearest-waypoint search: a figure-eight that crosses itself, a path with every waypoint twice (exact ties between
    neighbours), tables of 1 and 7 points, a line with a NaN waypoint, a closed circle whose last point equals its
    first (a tie between index 0 and the last index)."""
    Wmax = 203
    wp = np.zeros((6, Wmax, 2))
    wc = np.array([203, 200, 1, 7, 150, 181], dtype=np.int32)
    t = np.linspace(0.0, 2 * np.pi, 203)
    wp[0, :, 0], wp[0, :, 1] = 12.0 * np.sin(t), 6.0 * np.sin(2 * t)                   # figure-eight through the origin
    s = np.repeat(np.arange(100) * 0.05, 2)
    wp[1, :200, 0], wp[1, :200, 1] = s, 0.02 * s * s                                   # every point twice
    wp[2, 0] = (3.0, -1.0)
    wp[3, :7, 0], wp[3, :7, 1] = np.arange(7) * 0.4, 0.1 * np.arange(7)
    wp[4, :150, 0], wp[4, :150, 1] = np.arange(150) * 0.06, 1.0
    wp[4, 3] = (np.nan, np.nan)                                                        # never the nearest; behind every vehicle
    a = np.linspace(0.0, 2 * np.pi, 181)
    wp[5, :181, 0], wp[5, :181, 1] = 5.0 * np.cos(a), 5.0 * np.sin(a)
    wp[5, 180] = wp[5, 0]                      # last == first, bit for bit (sin(2 pi) is -2.4e-16, which would make the
                                               # choice hang on the last bit of x*x + y*y: fma or not, BLAS or not)
    wp[:, :, :][np.arange(6)[:, None] * 0 + np.arange(Wmax)[None, :] >= wc[:, None]] = 1e6   # garbage past each table's end
    return wp, wc


def test_closed_loop_search_on_adversarial_tables(gpu_vm, pkg, oracle):
    """The closed loop's LDS search (two levels of bounding circles, per-lane masks, packed block scan, hinted
    bound) on tables built to defeat it, 40 sub-steps = 4 controller updates (the first without a hint), vehicles on,
    near, and 200 m away from their paths (the bound is then so loose that whole tables survive the circles): fp64
    target indices and states exactly as the oracle's; chained launches == one launch; tables too big for LDS (the
    global-memory search) give the same bits; fp32 follows within rounding."""
    wp, wc = _adversarial_tables()
    rng = np.random.default_rng(99)
    n, dt, H = 1536, 1e-3, 40
    pid = (np.arange(n) % 6).astype(np.int32)
    k = (rng.uniform(0, 1, n) * (wc[pid] - 1)).astype(int)
    k[pid == 4] = np.maximum(k[pid == 4], 10)
    s0 = np.zeros((12, n))
    s0[0] = rng.uniform(8, 25, n)
    s0[3:7] = s0[0] / 0.308309813617345
    far = rng.uniform(0, 1, n) < 0.05
    s0[8] = np.nan_to_num(wp[pid, k, 0]) + rng.normal(0, 0.3, n) + 200.0 * far
    s0[9] = np.nan_to_num(wp[pid, k, 1]) + rng.normal(0, 0.3, n)
    s0[8, pid == 0] *= rng.uniform(0, 1, (pid == 0).sum()) < 0.5                      # half of the figure-eight's vehicles at the crossing
    s0[9, pid == 0] *= s0[8, pid == 0] != 0
    s0[7] = rng.uniform(-np.pi, np.pi, n)
    c0 = np.zeros((6, n))
    c0[2], c0[3] = s0[0], 25.0
    vm = gpu_vm(dt)
    cp = oracle.ctrl_params()
    term, cs, log = vm.closed_loop(s0, c0, wp, H, wcount=wc, path_id=pid, log=True)
    with np.errstate(all="ignore"):
        ot, oc, olog = oracle.closed_loop(oracle.default_params(), cp, s0, c0, wp, wc, pid, dt, H, log=True, nthreads=8)
    ok = np.isfinite(ot).all(axis=0)                                                  # the NaN table can make a walk non-finite in both
    assert ok.mean() > 0.8
    assert np.array_equal(log[::10, 14][:, ok], olog[::10, 14][:, ok]), "target indices must match the oracle exactly"
    from conftest import parity
    assert parity(term[:, ok], ot[:, ok], 1e-6) <= 1e-8
    a, ca = vm.closed_loop(s0, c0, wp, 13, wcount=wc, path_id=pid)
    b, cb = vm.closed_loop(a, ca, wp, 27, wcount=wc, path_id=pid, phase=13)
    assert np.array_equal(b[:, ok], term[:, ok]) and np.array_equal(cb[:, ok], cs[:, ok])
    big = np.full((6, 30000, 2), 1e6)                                                  # 6 x 30000 x 24 B: no LDS image
    big[:, :203] = wp
    t2, c2 = vm.closed_loop(s0, c0, big, H, wcount=wc, path_id=pid)
    assert np.array_equal(t2[:, ok], term[:, ok]) and np.array_equal(c2[:, ok], cs[:, ok])
    # fp32: same tables (203 floats per row: the scalar staging path).  A vehicle AT the figure-eight's crossing takes
    # either branch (the plain-C float oracle differs from the fp64 one on 12 % of that table's updates), so fp32 is held
    # to the FLOAT oracle.  The NaN-waypoint table is part of the comparison: the fp32 lookahead works on the cumulative
    # arc length, which a NaN waypoint poisons for everything behind it -- such a table is walked segment by segment, as
    # the reference walks every table (round 3 excluded the table and asked callers for finite fp32 tables instead).
    with np.errstate(all="ignore"):
        _, _, olog32 = oracle.closed_loop(oracle.default_params(), cp, s0.astype(np.float32), c0.astype(np.float32),
                                          wp.astype(np.float32), wc, pid, dt, H, log=True, nthreads=8)
    t32, c32, log32 = vm.closed_loop(s0.astype(np.float32), c0.astype(np.float32), wp.astype(np.float32), H, wcount=wc,
                                     path_id=pid, log=True)
    # table 4 (the NaN waypoint) on its own: whatever the poisoned arc lengths make of the lookahead, the fp32 kernel must
    # stay in bounds and finite -- target index inside [0, wcount), steering command finite and within the clip
    t4 = pid == 4
    idx4 = log32[::10, 14][:, t4]
    assert np.isfinite(idx4).all() and (idx4 >= 0).all() and (idx4 < wc[4]).all(), "fp32 target index out of the NaN table's range"
    assert np.isfinite(log32[:, 12][:, t4]).all() and np.abs(log32[:, 12][:, t4]).max() <= np.deg2rad(30) + 1e-6
    assert np.isfinite(t32[:, t4]).all() and np.isfinite(c32[:, t4]).all(), "fp32 state on the NaN table must stay finite"
    # the same table through the single-update entry point (tables in global memory, cumulative arcs from
    # waypoint_cumsum_kernel: the other fp32 lookahead path): finite, in range, and the float oracle's index
    cs_u, out_u = vm.controller_update(s0.astype(np.float32), c0.astype(np.float32), wp.astype(np.float32), wcount=wc, path_id=pid)
    assert np.isfinite(out_u[:, t4]).all() and (out_u[1, t4] >= 0).all() and (out_u[1, t4] < wc[4]).all()
    far_off = 0
    for i in np.where(t4)[0]:
        _, jf, _ = oracle.stanley_control(cp, wp[4, :wc[4]].astype(np.float32), np.float32(s0[8, i]), np.float32(s0[9, i]),
                                          np.float32(s0[7, i]), np.float32(s0[0, i]), dtype=np.float32)
        far_off += abs(int(out_u[1, i]) - jf) > 1
    assert far_off <= 0.02 * t4.sum(), f"{far_off} of {int(t4.sum())} single updates on the NaN table miss the float oracle's index"
    ok32 = ok & np.isfinite(t32).all(axis=0)
    off4 = (np.abs(log32[::10, 14][:, t4] - olog32[::10, 14][:, t4]) > 1).mean()
    print(f"\n  NaN-waypoint table, fp32: finite, in range, indices more than one off the float oracle's on {off4:.2%}")
    assert off4 <= 0.02
    didx = np.abs(log32[::10, 14][:, ok32] - olog32[::10, 14][:, ok32])
    off = (didx > 1).mean()
    print(f"\n  adversarial tables: {ok.mean():.1%} finite in the oracle, fp32 indices more than one off the float oracle's on {off:.2%}")
    assert off <= 0.02


def test_fp32_zero_length_segment_heading_is_the_references(gpu_vm, oracle):
    """where the target waypoint's segment has zero length -- a closed path whose last waypoint
    repeats the first (wrap branch at the last index), a one-waypoint table, duplicated consecutive waypoints -- the
    reference's trajectory heading is arctan2(0, 0) = 0 and its heading error wrap(-yaw)
    (stanley_controller.py:109-123).  The fp32 law works on (cross, dot) of the segment in the vehicle's frame and
    must give the same, not the atan2(+-0, +-0) of a rotated zero vector: fp32 device == float oracle == fp64 oracle
    to rounding, at yaw angles in all four quadrants."""
    wp, wc = _adversarial_tables()
    rng = np.random.default_rng(7)
    n = 3 * 256
    pid = np.repeat(np.array([5, 2, 1], dtype=np.int32), 256)       # closed circle | one waypoint | every point twice
    s = np.zeros((12, n))
    s[0] = rng.uniform(4, 20, n)
    s[7] = rng.uniform(-np.pi, np.pi, n)
    s[7, ::4] = rng.choice([0.75 * np.pi, -0.75 * np.pi, 0.25 * np.pi, -0.25 * np.pi], n // 4)
    k5 = rng.integers(160, 181, 256)                                # the circle's last metres: the lookahead runs off its end
    s[8, :256], s[9, :256] = wp[5, k5, 0] + rng.normal(0, 0.05, 256), wp[5, k5, 1] + rng.normal(0, 0.05, 256)
    s[8, 256:512], s[9, 256:512] = wp[2, 0, 0] + rng.normal(0, 2.0, 256), wp[2, 0, 1] + rng.normal(0, 2.0, 256)
    k1 = rng.integers(0, 60, 256)
    s[8, 512:], s[9, 512:] = wp[1, k1, 0] + rng.normal(0, 0.05, 256), wp[1, k1, 1] + rng.normal(0, 0.05, 256)
    c = np.zeros((6, n))
    c[2], c[3] = s[0], 25.0
    vm = gpu_vm(1e-3)
    cp = oracle.ctrl_params()
    cs64, o64 = vm.controller_update(s, c, wp, wcount=wc, path_id=pid)
    s32, c32, wp32 = s.astype(np.float32), c.astype(np.float32), wp.astype(np.float32)
    cs32, o32 = vm.controller_update(s32, c32, wp32, wcount=wc, path_id=pid)
    zero_len = 0
    for i in range(n):
        tab = wp[pid[i], :wc[pid[i]]]
        d64, i64, _ = oracle.stanley_control(cp, tab, s[8, i], s[9, i], s[7, i], s[0, i])
        df, jf, _ = oracle.stanley_control(cp, tab.astype(np.float32), s32[8, i], s32[9, i], s32[7, i], s32[0, i], dtype=np.float32)
        assert i64 == o64[1, i] and abs(d64 - o64[0, i]) <= 1e-9
        nxt = tab[i64 + 1] if i64 + 1 < len(tab) else tab[0]
        degenerate = bool((nxt == tab[i64]).all())
        zero_len += degenerate
        if int(o32[1, i]) == jf == i64:                              # same target: the steering angle must agree
            assert abs(float(o32[0, i]) - float(df)) <= 2e-4, (i, pid[i], i64, degenerate, o32[0, i], df, d64, s[7, i])
            assert abs(float(o32[0, i]) - d64) <= 3e-4, (i, pid[i], i64, degenerate, o32[0, i], d64, s[7, i])
    assert zero_len >= 300, f"only {zero_len} vehicles target a zero-length segment"
    # the one-waypoint table on its own: every vehicle there targets a zero-length segment, and none is clipped away
    # from the difference (|wrap(-yaw)| beyond max_steer clips both to the same value; keep the unclipped ones)
    one = (pid == 2) & (np.abs(o64[0]) < np.deg2rad(30) - 1e-3)
    assert one.sum() >= 5 and np.abs(o32[0, one] - o64[0, one]).max() <= 3e-4


def test_row_writer_limit_is_an_argument_error_not_a_fault(gpu_vm, pkg):
    """Trajectory / log / DataLog rows are addressed with a 64-bit base and 32-bit offsets (RowWriter,
    vdyn_kernels.hip): a row beyond 2^31 bytes is refused with VDYN_ERR_ARG before anything is launched
    (include/vdyn.h).  Small real buffers, a large claimed n: the check must come first."""
    import ctypes as C
    import torch
    vm = gpu_vm(1e-3)
    h = vm.handle()
    L = pkg._lib
    buf = torch.zeros(4096, dtype=torch.float32, device="cuda:0")
    ibuf = torch.zeros(64, dtype=torch.int32, device="cuda:0")
    vp = lambda t: C.c_void_p(t.data_ptr())
    n_big = (1 << 31) // (12 * 4) + 1
    with pytest.raises(L.VdynError) as e:
        h.call("vdyn_rollout_f32_dev", n_big, 1, vp(buf), vp(buf), 2, L.VDYN_CTRL_PER_ROLLOUT, None, 0, 1e-3, None,
               vp(buf), vp(buf), 1, None)
    assert e.value.code == L.VDYN_ERR_ARG and "2^31" in str(e.value)
    n_big = (1 << 31) // (45 * 4) + 1
    g = L.default_ctrl_gains()
    with pytest.raises(L.VdynError) as e:
        h.call("vdyn_closed_loop_f32_dev", C.byref(g), n_big, 1, 10, 0, vp(buf), vp(buf), vp(buf), 8, vp(ibuf), vp(ibuf), 1,
               1e-3, vp(buf), vp(buf), None, vp(buf), None)
    assert e.value.code == L.VDYN_ERR_ARG and "2^31" in str(e.value)
