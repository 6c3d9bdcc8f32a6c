"""Rank entry of tests/test_bench_multirank.py: bench.py's own ``run`` -- sharding by whole egos,
exchange, barrier-bracketed timing, JSON -- on CPU ranks (gloo), with the per-rank compute
replaced by the oracle (test infrastructure; the product has no CPU path).  Started by bench.py's
own ``spawn_ranks``."""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402


class OracleCompute:
    backend = "gloo"

    def __init__(self, pkg, local_rank, lanes_per_rollout, dt):
        import time
        self.device = torch.device("cpu")
        self.dt, self.p, self._t = dt, O.default_params(), time.perf_counter
        self.lanes = lanes_per_rollout
        # VDYN_TEST_FLIP_RANK=r: rank r's compute returns one element one ulp off -- what a damaged transfer
        # looks like to the ranks that receive its block (test_bench_multirank's negative case)
        self.flip = os.environ.get("VDYN_TEST_FLIP_RANK") == os.environ.get("RANK", "0")

    def rollout(self, s0, tab, pid):
        if os.environ.get("VDYN_TEST_HANG_STAGE") == "headline":    # a launch that never returns (every rank)
            import time
            while True:
                time.sleep(3600)
        t = O.rollout(self.p, s0.numpy().astype(np.float64), tab.numpy().astype(np.float64), self.dt,
                      path_id=pid.numpy(), nthreads=1)
        t = t.astype(np.float32)
        if self.flip:
            t[3, t.shape[1] // 2] = np.nextafter(t[3, t.shape[1] // 2], np.float32(np.inf))
        return torch.from_numpy(t)

    def with_lanes(self, lanes_per_rollout):
        return self         # the oracle has one mapping; bench.py's `strong` section only needs the object

    def handle(self):
        # VDYN_TEST_FAKE_P2P: pretend there is a library handle, so that bench.py's `auto` takes the peer-copy branch
        # (FakePeerExchange below) -- set-up, self test, calibration, relaunch logic -- on CPU ranks
        return object() if os.environ.get("VDYN_TEST_FAKE_P2P") else None

    def sync(self):
        pass

    def mark(self):
        return self._t()

    @staticmethod
    def elapsed_s(a, b):
        return b - a


class FakePeerExchange:
    """Stand-in of distributed.PeerExchange for CPU ranks (same interface; a gloo all-gather moves the blocks), with
    the two failures no GPU box available to the build can produce, switched on by VDYN_TEST_P2P_FAULT:
    `hang_in_start` -- every rank's first push blocks forever (a peer copy that never completes);
    `raise_after_agree` -- rank 1 raises in try_create AFTER the ranks agreed that the set-up worked, i.e. at a point
    where its peers have already gone on to the next collective."""
    kind = "peer_copies"
    fallback_reason = None

    def __init__(self, sh, rows, like, handle):
        import importlib
        D = importlib.import_module("python-motionplanning_amd.distributed")
        self._x = D.AllGatherExchange(sh, rows, like)
        self.fault = os.environ.get("VDYN_TEST_P2P_FAULT")

    @classmethod
    def try_create(cls, sh, rows, like, handle, self_test=True):
        import torch.distributed as dist
        x = cls(sh, rows, like, handle)
        t = torch.tensor([1], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)          # the agree step
        if x.fault == "raise_after_agree" and sh.rank == 1:
            raise RuntimeError("hipIpcOpenMemHandle: stand-in failure after the agree step")
        return x, None

    def start(self, term):
        if self.fault == "hang_in_start":
            import time
            while True:
                time.sleep(3600)
        self._x.start(term)

    def wait(self):
        self._x.wait()

    def result(self):
        return self._x.result()

    def close(self):
        self._x.close()


if __name__ == "__main__":
    if os.environ.get("VDYN_TEST_FAKE_P2P"):
        import importlib
        importlib.import_module("python-motionplanning_amd.distributed").PeerExchange = FakePeerExchange
    bench.main(compute_factory=OracleCompute, script=os.path.abspath(__file__))
