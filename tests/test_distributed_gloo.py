"""CPU, world_size 2, gloo: the N > 1 path of the rollout batch -- whole-ego
sharding and the terminal-state all-gather -- with the oracle standing in for
the per-rank HIP compute (the product has no CPU path; tests may use the oracle)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, H, out_dir):
    sys.path.insert(0, REPO)
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pkg = importlib.import_module("python-motionplanning_amd")
        D = importlib.import_module("python-motionplanning_amd.distributed")
        from oracle import oracle as O
        W = pkg.workloads
        p = O.default_params()
        s0, tab, pid = W.config3(n, H, np.float64)
        sh = D.ShardedRollout(n)
        assert sh.lo % W.NUM_PATHS == 0, "shards start on an ego boundary"

        def local_rollout(state0_local, table, path_id):
            t = O.rollout(p, state0_local.numpy(), table, 1e-3, path_id=path_id)
            return torch.from_numpy(t)

        full = sh.rollout(local_rollout, torch.from_numpy(np.ascontiguousarray(sh.local(s0))), tab,
                          sh.local(pid))
        np.save(os.path.join(out_dir, f"gather_{rank}.npy"), full.numpy())
        # MPC: egos sharded, candidates replicated, per-ego argmin is rank-local
        E = 10
        ego, cand, goal = W.config5(E, 24, 10, np.float64)
        she = D.ShardedRollout(E, units_per_ego=1)
        bc, bi = O.mpc_argmin(p, she.local(ego), cand, she.local(goal), 2e-3, W.MPC_W_DELTA)
        c, i = D.all_gather_argmin(torch.from_numpy(bc), torch.from_numpy(bi), E)
        np.save(os.path.join(out_dir, f"mpc_{rank}.npy"), np.stack([c.numpy(), i.numpy().astype(np.float64)]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [70, 65])  # 65: ragged (last ego partial, ranks unequal)
def test_sharded_rollout_equals_single_process(tmp_path, n, oracle, workloads):
    world, H = 2, 25
    mp.spawn(_worker, args=(world, _free_port(), n, H, str(tmp_path)), nprocs=world, join=True)
    s0, tab, pid = workloads.config3(n, H, np.float64)
    p = oracle.default_params()
    single = oracle.rollout(p, s0, tab, 1e-3, path_id=pid)
    for r in range(world):
        got = np.load(tmp_path / f"gather_{r}.npy")
        assert np.array_equal(got, single), "shard + all-gather must be bitwise the single-rank result"
    ego, cand, goal = workloads.config5(10, 24, 10, np.float64)
    bc, bi = oracle.mpc_argmin(p, ego, cand, goal, 2e-3, workloads.MPC_W_DELTA)
    for r in range(world):
        m = np.load(tmp_path / f"mpc_{r}.npy")
        assert np.array_equal(m[0], bc) and np.array_equal(m[1].astype(np.int32), bi)


def test_shard_bounds_cover_everything(workloads):
    for n in (0, 1, 6, 7, 8, 65534, 65536):
        for world in (1, 2, 3, 8):
            b = [workloads.shard_egos(n, world, r) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            for (lo0, hi0), (lo1, hi1) in zip(b, b[1:]):
                assert hi0 == lo1 and lo0 <= hi0
            assert all(lo % workloads.NUM_PATHS == 0 or lo == n for lo, _ in b)  # empty tail shards sit at n


def test_peer_exchange_slot_arithmetic(pkg):
    """Byte geometry of distributed.PeerExchange's slot buffers (the copies themselves need GPUs)."""
    import importlib
    D = importlib.import_module("python-motionplanning_amd.distributed")
    block, total, offs = D.slot_layout(8, 12, 65534, 4)
    assert block == 12 * 65534 * 4 and total == 8 * block and offs == [r * block for r in range(8)]
    assert D.slot_layout(1, 12, 7, 8) == (672, 672, [0])
    # slots tile the buffer exactly, in rank order, and match what assemble() reads back
    import torch
    sh = D.ShardedRollout(65)                       # one process: world 1
    sh.world, sh.bounds = 3, [(0, 28), (28, 56), (56, 65)]
    sh.n_pad = 28
    flat = torch.zeros(3 * 12 * 28)
    _, _, offs = D.slot_layout(3, 12, 28, 1)
    for r, (lo, hi) in enumerate(sh.bounds):        # what rank r pushes: its padded block at its slot offset
        blk = torch.zeros(12, 28)
        blk[:, :hi - lo] = torch.arange(lo, hi, dtype=torch.float32)[None, :] + 1000.0 * torch.arange(12)[:, None]
        flat[offs[r]:offs[r] + 12 * 28] = blk.reshape(-1)
    full = sh.assemble(flat.view(3 * 12, 28), 12)
    want = torch.arange(65, dtype=torch.float32)[None, :] + 1000.0 * torch.arange(12)[:, None]
    assert torch.equal(full, want)


class _MockXchgHandle:
    """Stands in for the library handle in PeerExchange's set-up: `vdyn_xchg_alloc` hands out host memory (or raises,
    on the rank told to fail), the other entry points record that they were called."""

    def __init__(self, fail_alloc):
        self.fail_alloc, self.calls, self.buf = fail_alloc, [], None

    def call(self, name, *args):
        import ctypes as C
        self.calls.append(name)
        if name == "vdyn_xchg_alloc":
            if self.fail_alloc:
                raise RuntimeError("mock: hipMalloc failed on this rank")
            total, own_ref = args[0], args[1]
            self.buf = (C.c_char * int(total))()
            own_ref._obj.value = C.addressof(self.buf)
        elif name == "vdyn_xchg_open":
            raise RuntimeError("mock: no peer mapping on the CPU")


def _peer_failure_worker(rank, world, port, out_dir):
    sys.path.insert(0, REPO)
    import ctypes as C
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        importlib.import_module("python-motionplanning_amd")
        D = importlib.import_module("python-motionplanning_amd.distributed")

        def host_buffer(ptr, shape, typestr):       # the slot buffer as a NumPy view of the mock's host memory
            n = int(np.prod(shape))
            return np.ctypeslib.as_array((C.c_float * n).from_address(int(ptr))).reshape(shape)
        D._DeviceBuffer = host_buffer
        torch.cuda.synchronize = lambda *a, **k: None
        sh = D.ShardedRollout(70)
        like = torch.zeros((12, sh.n_local), dtype=torch.float32)
        h = _MockXchgHandle(fail_alloc=(rank == 1))
        x, why = D.PeerExchange.try_create(sh, 12, like, h)
        assert x is None and why, (x, why)
        # the very next collective of the `auto` fallback: must pair up on every rank (the rank that had nothing to
        # free used to skip close()'s barrier, and its peers' barrier then met THIS all-gather)
        fb = D.make_exchange("rccl", sh, 12, like)
        fb.start(torch.full((12, sh.n_local), float(rank + 1)))
        got = fb.result()
        for r, (lo, hi) in enumerate(sh.bounds):
            assert bool((got[:, lo:hi] == float(r + 1)).all())
        with open(os.path.join(out_dir, f"peer_failure_{rank}.txt"), "w") as f:
            f.write(why + "\n" + ",".join(h.calls))
    finally:
        dist.destroy_process_group()


def test_peer_exchange_one_rank_cannot_allocate_all_ranks_fall_back_together(tmp_path):
    """PeerExchange.close() is collective, also for the rank that holds nothing.  Rank 1's
    vdyn_xchg_alloc raises; both ranks must get (None, reason) from try_create and then complete an all-gather."""
    world = 2
    ctx = mp.spawn(_peer_failure_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 120:
            for pr in ctx.processes:
                pr.kill()
            pytest.fail("the ranks did not come back: a barrier met another collective")
    why0 = (tmp_path / "peer_failure_0.txt").read_text()
    why1 = (tmp_path / "peer_failure_1.txt").read_text()
    assert "could not export" in why0 or "another rank" in why0
    assert "rank 1" in why1 and "mock: hipMalloc failed" in why1
    assert "vdyn_xchg_free" in why0.splitlines()[1], "the healthy rank frees what it allocated"
