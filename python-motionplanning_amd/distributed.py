"""Multi-GPU sharding of a rollout batch: one process per GPU, ``torch.distributed``
(backend ``nccl`` = RCCL over xGMI on MI355X; ``gloo`` for the CPU rehearsal in tests).

Rollouts are independent (no coupling between vehicles anywhere in
vehicle_model.py:220-445; the RK4 recurrence is per rollout), so the batch is
split into contiguous blocks of WHOLE egos -- all lattice paths / MPC candidates
of an ego stay on one rank -- and the only exchange the path has is one
all-gather of the terminal states (``[12][n_local]`` per rank, 393 KB at
8192 fp32 rollouts) or of the per-ego ``(best_cost, best_idx)`` pairs.
There is no collective inside the horizon.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .workloads import NUM_PATHS, shard_egos


class ShardedRollout:
    """Shard ``n_units`` rollouts over the ranks of ``group`` by whole egos.

    ``rollout_fn(state0_local, *args_local) -> terminal_local [12][n_local]`` is the
    per-rank compute -- ``VehicleModel.rollout`` on a GPU rank."""

    def __init__(self, n_units, group=None, units_per_ego=NUM_PATHS):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n_units = int(n_units)
        self.units_per_ego = int(units_per_ego)
        self.bounds = [shard_egos(self.n_units, self.world, r, self.units_per_ego)
                       for r in range(self.world)]
        self.lo, self.hi = self.bounds[self.rank]
        self.n_local = self.hi - self.lo
        self.n_pad = max(hi - lo for lo, hi in self.bounds)  # equal-size all-gather blocks

    def local(self, x, axis=-1):
        """This rank's slice of a global array/tensor along the rollout axis."""
        idx = [slice(None)] * x.ndim
        idx[axis] = slice(self.lo, self.hi)
        return x[tuple(idx)]

    def assemble(self, blocks: torch.Tensor, rows: int) -> torch.Tensor:
        """Rank-major padded blocks ``[world][rows][n_pad]`` -> ``[rows][n_units]``."""
        blocks = blocks.view(self.world, rows, self.n_pad)
        out = blocks.new_empty((rows, self.n_units))
        for r, (lo, hi) in enumerate(self.bounds):
            out[:, lo:hi] = blocks[r, :, :hi - lo]
        return out

    def all_gather_terminal(self, term_local: torch.Tensor) -> torch.Tensor:
        """[rows][n_local] on every rank -> [rows][n_units] on every rank."""
        if self.world == 1:
            return term_local
        x = AllGatherExchange(self, term_local.shape[0], term_local)
        x.start(term_local)
        return x.result()

    def rollout(self, rollout_fn, state0_local, *args, **kw):
        """Local compute + the terminal-state all-gather."""
        return self.all_gather_terminal(rollout_fn(state0_local, *args, **kw))


class AllGatherExchange:
    """The one exchange step of the path -- every rank ends up with every rank's terminal block --
    as ``torch.distributed.all_gather_into_tensor`` (RCCL over xGMI on GPU ranks, gloo on CPU
    ranks) into a preallocated buffer of rank-major, equally sized blocks ``[world][rows][n_pad]``.

    ``start`` is asynchronous: the collective runs beside the caller's stream, ``wait`` joins it
    (bench.py queues the next rollout in between, so exchange k overlaps rollout k + 1)."""

    kind = "all_gather_into_tensor"

    def __init__(self, sh: ShardedRollout, rows: int, like: torch.Tensor):
        self.sh, self.rows = sh, int(rows)
        self.recv = like.new_empty((sh.world * self.rows, sh.n_pad))
        # a rank holding fewer egos than the largest shard sends a zero-padded block
        self.send = like.new_zeros((self.rows, sh.n_pad)) if sh.n_local != sh.n_pad else None
        self._pending = None

    def start(self, term_local: torch.Tensor):
        assert self._pending is None, "wait() for the previous exchange first"
        assert tuple(term_local.shape) == (self.rows, self.sh.n_local)
        src = term_local
        if self.send is not None:
            self.send[:, :self.sh.n_local].copy_(term_local)
            src = self.send
        if not dist.is_initialized():
            self.recv.copy_(src)
            return
        work = dist.all_gather_into_tensor(self.recv, src.contiguous(), group=self.sh.group, async_op=True)
        self._pending = (work, src)          # the source stays referenced until the collective is joined

    def wait(self):
        if self._pending is not None:
            self._pending[0].wait()
            self._pending = None

    def result(self) -> torch.Tensor:
        """[rows][n_units]: the blocks of the last exchange, unpadded and in global order."""
        self.wait()
        return self.sh.assemble(self.recv, self.rows)

    def close(self):
        self.wait()


def make_exchange(kind: str, sh: ShardedRollout, rows: int, like: torch.Tensor):
    if kind == "rccl":
        return AllGatherExchange(sh, rows, like)
    raise ValueError(f"unknown exchange {kind!r}")


def all_gather_argmin(best_cost_local: torch.Tensor, best_idx_local: torch.Tensor, n_ego, group=None):
    """MPC (config 5): egos are sharded, candidates are replicated, so the per-ego
    argmin is rank-local; gather the (cost, idx) pairs, 8 B per ego."""
    sh = ShardedRollout(n_ego, group, units_per_ego=1)
    c = sh.all_gather_terminal(best_cost_local[None, :])[0]
    i = sh.all_gather_terminal(best_idx_local[None, :])[0]
    return c, i
