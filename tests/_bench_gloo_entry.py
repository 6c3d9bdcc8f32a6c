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
        t = O.rollout(self.p, s0.numpy().astype(np.float64), tab.numpy().astype(np.float64), self.dt,
                      path_id=pid.numpy(), nthreads=1)
        t = t.astype(np.float32)
        if self.flip:
            t[3, t.shape[1] // 2] = np.nextafter(t[3, t.shape[1] // 2], np.float32(np.inf))
        return torch.from_numpy(t)

    def with_lanes(self, lanes_per_rollout):
        return self         # the oracle has one mapping; bench.py's `strong` section only needs the object

    def handle(self):
        return None

    def sync(self):
        pass

    def mark(self):
        return self._t()

    @staticmethod
    def elapsed_s(a, b):
        return b - a


if __name__ == "__main__":
    args = bench.parse()
    bench.run(args, compute_factory=OracleCompute)
