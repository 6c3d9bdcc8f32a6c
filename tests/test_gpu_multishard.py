"""GPU (-m gpu): the multi-GPU contract checked on ONE MI355X, and BASELINE configs[4] at its
full size.

SURVEY.md section 8(e): ranks own contiguous blocks of whole egos, run the same lane-per-rollout
kernel, and the gathered result must be bit for bit the single-GPU result.  An 8-GPU node is not
available to the tests, so the 8 shards `workloads.shard_egos(65536, 8, r)` are integrated one
after another on cuda:0 and their concatenation is compared with one 65536-rollout launch."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_eight_shards_one_after_another_equal_the_single_launch_bitwise(gpu_vm, workloads, dtype):
    import torch
    dev = torch.device("cuda:0")
    n, H, world = 65536, 200, 8
    s0, tab, pid = workloads.config3(n, H, dtype)
    vm = gpu_vm(1e-3)                                   # lanes_per_rollout = 1: what bench.py runs on every shard
    s0d, tabd, pidd = (torch.from_numpy(a).to(dev) for a in (s0, tab, pid))
    single = vm.rollout(s0d, tabd, path_id=pidd)
    parts, covered = [], 0
    for r in range(world):
        lo, hi = workloads.shard_egos(n, world, r)
        assert lo == covered and lo % workloads.NUM_PATHS == 0
        covered = hi
        parts.append(vm.rollout(s0d[:, lo:hi].contiguous(), tabd, path_id=pidd[lo:hi].contiguous()))
    assert covered == n
    torch.cuda.synchronize()
    assert torch.equal(torch.cat(parts, dim=1), single), "shard + concatenate must be bitwise the single launch"
    # the weak-scaling shard of bench.py (9362 whole egos = 65534 rollouts) is a prefix of the same batch
    lo, hi = workloads.shard_egos(8 * 65534, 8, 0)
    assert (lo, hi) == (0, 65534)
    assert torch.equal(vm.rollout(s0d[:, :hi].contiguous(), tabd, path_id=pidd[:hi].contiguous()), single[:, :hi])


def test_mpc_config5_full_size_vs_oracle(gpu_vm, oracle, workloads):
    """BASELINE configs[4] as worded: 1024 egos x 512 candidates x 50 steps, dt = 2e-3, argmin on
    the device.  Cost model after collision_checker.py:175,190-196 (terminal distance to the goal;
    non-finite candidates disqualified; strict '<' scan, lowest index on ties)."""
    E, C, H, dt = 1024, 512, 50, 2e-3
    ego, cand, goal = workloads.config5(E, C, H)                      # fp32, the bench workload
    vm = gpu_vm(dt)
    bc, bi, cost = vm.mpc_argmin(ego, cand, goal, w_delta=workloads.MPC_W_DELTA, return_costs=True)
    e64, c64, g64 = (a.astype(np.float64) for a in (ego, cand, goal))
    p = oracle.default_params()
    obc, obi, ocost = oracle.mpc_argmin(p, e64, c64, g64, dt, workloads.MPC_W_DELTA,
                                        nthreads=oracle.max_threads(), return_costs=True)
    assert np.isfinite(cost).all() and cost.shape == (E, C)
    worst = np.abs(cost - ocost).max()
    assert worst <= 1e-3, f"fp32 costs vs fp64 oracle: {worst:.3e}"
    assert np.array_equal(bi, cost.argmin(axis=1)), "device argmin == argmin of its own costs (first minimum)"
    assert np.array_equal(bc, cost.min(axis=1))
    rows = np.arange(E)
    gap = ocost[rows, bi] - obc                                        # the winner's true cost above the true minimum
    assert (gap >= 0).all() and gap.max() <= 2 * worst + 1e-6, f"winner not a minimiser up to fp32 noise: {gap.max():.3e}"
    print(f"\n  config5 full: fp32 cost err {worst:.2e}, argmin agrees with fp64 oracle on {(bi == obi).mean():.1%}, "
          f"worst winner gap {gap.max():.2e}")
    # fp64 on the device: the very same argmin as the oracle
    bc64, bi64, cost64 = vm.mpc_argmin(e64, c64, g64, w_delta=workloads.MPC_W_DELTA, return_costs=True)
    assert np.abs(cost64 - ocost).max() <= 1e-9
    assert np.array_equal(bi64, obi)
    assert np.abs(bc64 - obc).max() <= 1e-9


@pytest.mark.parametrize("n", [70 * 64, 65 * 64 + 3])          # equal shards; ragged (padded blocks, partial last ego)
def test_peer_copy_exchange_two_processes_one_gpu(tmp_path, gpu_vm, workloads, n):
    """distributed.PeerExchange (the exchange that leaves the CUs alone: IPC-exported slot buffers,
    device-to-device copies on a copy stream) with two processes on the one GPU of the box: every rank
    must end up with all terminal blocks, bit for bit the single launch."""
    import os
    import subprocess
    import sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H = 40
    entry = os.path.join(repo, "tests", "_peer_exchange_entry.py")
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.spawn_ranks(2, %r, script=%r))" % (repo, [str(tmp_path), str(n), str(H)], entry))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    s0, tab, pid = workloads.config3(n, H, np.float32)
    dev = torch.device("cuda:0")
    single = gpu_vm(1e-3).rollout(torch.from_numpy(s0).to(dev), torch.from_numpy(tab).to(dev),
                                  path_id=torch.from_numpy(pid).to(dev)).cpu().numpy()
    for r in range(2):
        got = np.load(tmp_path / f"p2p_rank{r}.npy")
        assert got.shape == single.shape and np.array_equal(got, single)


@pytest.mark.parametrize("mode,exchange", [("weak", "p2p"), ("strong", "p2p"), ("weak", "rccl"), ("weak", "auto")])
def test_bench_two_ranks_on_one_gpu_end_to_end(tmp_path, gpu_vm, workloads, mode, exchange):
    """bench.py's WHOLE N > 1 path on hardware, as far as one GPU allows: `python bench.py --gpus 2` starts its two
    ranks itself, both ranks integrate their whole-ego shard with the HIP kernel on GPU 0, the terminal blocks travel
    through the exchange (peer copies, or all_gather_into_tensor on device tensors), timing is barrier-bracketed, rank 0
    prints the JSON line.  (RCCL refuses two ranks on one device, so torch.distributed runs on gloo here -- the
    all-gather then stages through the host, which only its speed notices; RCCL itself is rehearsed with
    --force-collective.)"""
    import json
    import os
    import subprocess
    import sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    per_gpu = 7 * 1000
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--prewarm-ms", "0",
           "--no-extra", "--no-cpu-baseline", "--device-map", "0,0", "--dist-backend", "gloo", "--exchange", exchange,
           "--rollouts-per-gpu", str(per_gpu), "--horizon", "50", "--dump-gathered", str(tmp_path)]
    if mode == "strong":
        cmd.append("--strong")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    n_total = 2 * per_gpu if mode == "weak" else per_gpu
    assert out["n_gpus"] == 2 and out["world_seen"] == 2 and out["rollouts_total"] == n_total and out["scaling"] == mode
    assert out["shards"] == [list(workloads.shard_egos(n_total, 2, r)) for r in range(2)]
    assert "bitwise" in out["exchange"].pop("verified_how")
    # `auto` MEASURES which exchange the timed region uses (exchange_calibration): whichever it chose must be what ran
    kind = out["exchange_calibration"]["chosen"] if exchange == "auto" else \
        ("peer_copies" if exchange == "p2p" else "all_gather_into_tensor")
    why = out["exchange"].pop("fallback_reason")
    assert (why is None) if (exchange != "auto" or kind == "peer_copies") else why.startswith("calibration:")
    assert out["exchange"] == {"kind": kind, "overlapped": True, "bytes_per_rank": 12 * (out["shards"][0][1]) * 4,
                               "verified": True, "requested": exchange, "peer_copies_disabled": False}
    assert out["attempt"] == 1 and "relaunched" not in out and "timed_out_in" not in out
    assert out["value"] > 0 and out["roofline"]["bound"] == "valu" and "cpu_baseline" not in out
    cal = out["exchange_calibration"]
    if exchange == "auto":      # measured before the timed region: peer copies against the all-gather, same on every rank
        assert cal["chosen"] == out["exchange"]["kind"] and cal["peer_copies_ms_per_step"] > 0 and cal["all_gather_ms_per_step"] > 0
    else:
        assert cal is None
    s0, tab, pid = workloads.config3(n_total, 50, np.float32)
    dev = torch.device("cuda:0")
    single = gpu_vm(1e-3).rollout(torch.from_numpy(s0).to(dev), torch.from_numpy(tab).to(dev),
                                  path_id=torch.from_numpy(pid).to(dev)).cpu().numpy()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"gathered_rank{r}.npy"), single), "what every rank holds == the single launch"


@pytest.mark.parametrize("exchange,overlap", [("rccl", True), ("rccl", False), ("p2p", True), ("auto", True)])
def test_bench_nccl_backend_one_rank_force_collective(exchange, overlap):
    """The DEFAULT multi-GPU path of bench.py as an 8-GPU run executes it -- torch.distributed on the `nccl` backend
    (= RCCL) with `device_id` set, `all_gather_into_tensor(async_op=True)` overlapped with the next launch
    (distributed.AllGatherExchange), or the peer-copy exchange on the same backend -- rehearsed with ONE rank on the
    one GPU of the box (`--force-collective`; RCCL refuses two ranks on one device, hence one).  A fresh child
    process: nothing here re-executes a process that touched the GPU."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--gpus", "1", "--force-collective", "--exchange", exchange,
           "--steps", "3", "--warmup", "1", "--prewarm-ms", "0", "--no-extra", "--no-cpu-baseline"]
    if not overlap:
        cmd.append("--no-overlap")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["world_seen"] == 1 and out["rollouts_total"] == 65536
    assert out["dist_backend"] == "nccl"
    assert "bitwise" in out["exchange"].pop("verified_how")
    kind = out["exchange_calibration"]["chosen"] if exchange == "auto" else \
        ("all_gather_into_tensor" if exchange == "rccl" else "peer_copies")
    why = out["exchange"].pop("fallback_reason")
    assert (why is None) if (exchange != "auto" or kind == "peer_copies") else why.startswith("calibration:")
    assert out["exchange"] == {"kind": kind, "overlapped": overlap, "bytes_per_rank": 12 * 65536 * 4, "verified": True,
                               "requested": exchange, "peer_copies_disabled": False}
    assert np.isfinite(out["value"]) and out["value"] > 0 and np.isfinite(out["ms_per_step"])
    assert out["roofline"]["bound"] == "valu" and out["shards"] == [[0, 65536]]
    assert (out["exchange_calibration"] is not None) == (exchange == "auto")
    # the sections an 8-GPU run adds to its one line: RCCL and peer copies side by side, the fixed-65536 split with both
    # kernels -- each verified -- and the per-GPU HBM roofline
    ab, st = out["exchange_ab"], out["strong"]
    for kind, name in (("rccl", "all_gather_into_tensor"), ("p2p", "peer_copies")):
        assert ab[kind]["available"] and ab[kind]["kind"] == name and ab[kind]["verified"] is True, ab
        assert 0 < ab[kind]["ms_per_step"] < 5.0
    assert st["rollouts_total"] == 65536 and st["shards"] == [[0, 65536]] and "error" not in st
    assert st["lane"]["verified"] is True and st["wheel_parallel"]["verified"] is True
    assert st["lane"]["lanes_per_rollout"] == 1 and st["wheel_parallel"]["lanes_per_rollout"] == 4
    assert 0 < out["roofline_hbm"]["frac"] < 1 and out["roofline_hbm"]["per_gpu"] is True
    assert "sections_timed_out" not in out


def test_seven_destinations_per_push_through_seven_copy_streams(gpu_vm, pkg):
    """The push of an EIGHT-rank job (n_dst = 7: one hipMemcpyAsync per peer, each on its own copy stream behind one
    event) had never executed in any form -- the one-GPU box admits six GPU processes.  Here the seven destinations are
    seven slot buffers of the same process (what differs from a real peer is only how the pointer was obtained): sixteen
    pushes back to back from the compute stream, a device-side fence every fourth, sources overwritten as soon as the
    fence allows -- the steady state of distributed.PeerExchange.start -- then the last block must be the one in every
    slot, and the host time per push (the 8 us-per-peer projection of DESIGN.md section 6) is printed."""
    import ctypes as C
    import time
    import torch
    L = pkg._lib
    vm = gpu_vm(1e-3)
    h = vm.handle()
    dev = torch.device("cuda:0")
    world, rank, rows, n_pad = 8, 3, 12, 8192
    block = rows * n_pad * 4
    own, ipc = [C.c_void_p() for _ in range(world)], [L.VdynIpcHandle() for _ in range(world)]
    for r in range(world):
        h.call("vdyn_xchg_alloc", world * block, C.byref(own[r]), C.byref(ipc[r]))
    dst = (C.c_void_p * 7)(*[own[r].value for r in range(world) if r != rank])
    try:
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ring = [torch.empty((rows, n_pad), dtype=torch.float32, device=dev) for _ in range(8)]   # 2 * HOLD sources
        # one untimed push first: it creates the seven copy streams and their events
        h.call("vdyn_xchg_push", dst, 7, rank * block, C.c_void_p(ring[0].data_ptr()), block, stream)
        h.call("vdyn_xchg_wait")
        torch.cuda.synchronize()
        t_host, t_all = 0.0, time.perf_counter()
        for i in range(16):
            if i >= 8 and i % 4 == 0:
                h.call("vdyn_xchg_fence", stream)            # the four oldest sources may be rewritten behind it
            src = ring[i % 8]
            src.fill_(float(i + 1))                          # on the compute stream, ahead of the push
            t0 = time.perf_counter()
            h.call("vdyn_xchg_push", dst, 7, rank * block, C.c_void_p(src.data_ptr()), block, stream)
            t_host += time.perf_counter() - t0
        t_enq = time.perf_counter() - t_all
        h.call("vdyn_xchg_wait")
        torch.cuda.synchronize()
        t_done = time.perf_counter() - t_all
        for r in range(world):
            if r == rank:
                continue
            slots = torch.as_tensor(pkg.distributed._DeviceBuffer(own[r].value, (world, rows * n_pad), "<f4"), device=dev)
            assert bool((slots[rank] == 16.0).all()), f"slot {rank} of buffer {r} does not hold the last block"
        print(f"\n  n_dst = 7: host time per push {t_host / 16 * 1e6:.1f} us ({t_host / 16 / 7 * 1e6:.1f} us per destination); "
              f"sixteen pushes of seven 393 KB blocks queued in {t_enq * 1e3:.2f} ms, landed after {t_done * 1e3:.2f} ms")
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "seven_destinations_push.txt"), "w") as fh:
                fh.write(f"vdyn_xchg_push, n_dst = 7 (seven slot buffers of one process on one MI355X, seven copy streams), "
                         f"16 pushes of 7 x {block} B: host {t_host / 16 * 1e6:.1f} us per push = {t_host / 16 / 7 * 1e6:.1f} us per "
                         f"destination; queued in {t_enq * 1e3:.3f} ms, all landed after {t_done * 1e3:.3f} ms\n")
    finally:
        h.call("vdyn_xchg_wait")
        for r in range(world):
            h.call("vdyn_xchg_free", own[r])
