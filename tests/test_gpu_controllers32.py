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
