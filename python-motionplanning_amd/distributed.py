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
    fallback_reason = None

    def __init__(self, sh: ShardedRollout, rows: int, like: torch.Tensor):
        self.sh, self.rows = sh, int(rows)
        self.recv = like.new_empty((sh.world * self.rows, sh.n_pad))
        # a rank holding fewer egos than the largest shard sends a zero-padded block
        self.send = like.new_zeros((self.rows, sh.n_pad)) if sh.n_local != sh.n_pad else None
        self._pending = None

    def start(self, term_local: torch.Tensor):
        # one receive buffer: the previous collective is joined first (for RCCL a stream-level wait: the compute
        # stream waits, the host does not)
        self.wait()
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


def slot_layout(world: int, rows: int, n_pad: int, itemsize: int):
    """Byte geometry of a rank's slot buffer ``[world][rows][n_pad]``: (block_bytes, total_bytes,
    [byte offset of rank r's slot])."""
    block = int(rows) * int(n_pad) * int(itemsize)
    return block, block * int(world), [r * block for r in range(int(world))]


class _DeviceBuffer:
    """A raw device pointer as something ``torch.as_tensor`` understands."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2}


class PeerExchange:
    """The same exchange without a collective kernel: every rank owns a slot buffer
    ``[world][rows][n_pad]`` (exported once over IPC), and per step pushes its block into slot
    ``rank`` of every rank's buffer with plain device-to-device copies on the library's copy
    stream (``vdyn_xchg_*``, include/vdyn.h) -- SDMA / xGMI traffic that leaves the CUs to the next
    rollout.  One process per GPU of ONE node.

    ``start`` orders the copies behind the caller's current stream and returns: the HOST never waits in the steady
    state.  Exchanges follow each other on the device -- a destination's copies run on one in-order copy stream, so a
    later block lands after an earlier one -- and the one thing that needs protecting, the SOURCE block until the
    copies have read it, is protected on the device as well: the last ``2 * HOLD`` sources stay referenced (so the
    caching allocator cannot hand their memory to the next rollout), and every ``HOLD`` exchanges ONE device-side wait
    (``vdyn_xchg_fence``: the caller's stream waits for the copies queued so far -- the newest of them belongs to the
    rollout before the one just launched and finishes well inside it) releases the ``HOLD`` oldest -- a stream-wait is
    a barrier packet between two kernels, so it is paid once per ``HOLD`` steps, not per step.  A padded send block
    rotates through ``2 * HOLD`` buffers under the same fence.  A rank's OWN block is not copied per step at all:
    ``result`` places the last one in its slot.
    ``wait`` blocks the host until THIS rank's copies have landed; a block pushed by another rank is known to have
    landed once that rank waited and both passed a barrier -- ``result`` and bench.py's fence do exactly that.
    UNMEASURED on more than one GPU (none was available to the build); the 2- and 4-process rehearsals on one GPU
    exercise handles, slots and ordering."""

    HOLD = 4            # 2 * HOLD sources kept alive; one fence per HOLD exchanges
    kind = "peer_copies"
    fallback_reason = None

    def __init__(self, sh: ShardedRollout, rows: int, like: torch.Tensor, handle):
        x, why = self._build(self, sh, rows, like, handle)
        if x is None:
            raise RuntimeError("peer-copy exchange unavailable: " + why)

    @classmethod
    def try_create(cls, sh: ShardedRollout, rows: int, like: torch.Tensor, handle, self_test=True):
        """(exchange, None), or (None, reason) when ANY rank could not set it up or the self test -- one real
        exchange of a known pattern, checked on every rank -- failed.  Collective: every rank calls it, every rank
        gets the same verdict, and no rank is left waiting in a collective for one that gave up (local failures are
        recorded and agreed on with an all-reduce before the next collective step)."""
        obj = cls.__new__(cls)
        x, why = cls._build(obj, sh, rows, like, handle)
        if x is None:
            return None, why
        if self_test:
            # in two agreed stages, so that a rank whose pushes fail never leaves the others inside result()'s barriers:
            # (1) push and wait for the own copies -- no collective inside; (2) the collective read-back and the check
            ok = True
            try:
                src = like.new_full((x.rows, sh.n_local), float(sh.rank + 1))
                x.start(src)
                x.wait()
            except Exception as e:                      # noqa: BLE001 -- any failure means "do not use"
                ok, why = False, f"self test: pushing to the peers raised {e!r}"
                x._inflight = []
            if not x._agree(ok):
                x.close()
                return None, why or "self test: a rank could not push its block to its peers"
            try:
                got = x.result()
                for r, (lo, hi) in enumerate(sh.bounds):
                    ok = ok and bool((got[:, lo:hi] == float(r + 1)).all())
            except Exception as e:                      # noqa: BLE001
                ok, why = False, f"self test: reading the slots back raised {e!r}"
            if not x._agree(ok):
                x.close()
                return None, why or "self test: a rank did not receive every rank's block"
        return x, None

    def _agree(self, ok: bool) -> bool:
        """Logical AND of `ok` over the ranks (all-reduce MIN on the group's own backend)."""
        if not (dist.is_initialized() and self.sh.world > 1):
            return bool(ok)
        on_gpu = "nccl" in str(dist.get_backend(self.sh.group))
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self._device if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.sh.group)
        return bool(int(t.item()) == 1)

    @staticmethod
    def _build(self, sh, rows, like, handle):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        self.sh, self.rows, self.h = sh, int(rows), handle
        self._device = like.device
        self.itemsize = like.element_size()
        self.block, total, self.offsets = slot_layout(sh.world, rows, sh.n_pad, self.itemsize)
        self._own, self._peers, self._inflight, self._count, self.recv, self._closed = None, [], [], 0, None, False
        self._last = None
        why = None
        own, ipc = C.c_void_p(), _lib.VdynIpcHandle()
        try:
            handle.call("vdyn_xchg_alloc", total, C.byref(own), C.byref(ipc))
            self._own = own.value
            typestr = {4: "<f4", 8: "<f8"}[self.itemsize]
            self.recv = torch.as_tensor(_DeviceBuffer(self._own, (sh.world * self.rows, sh.n_pad), typestr),
                                        device=like.device)
            self.recv.zero_()
            # a rank holding fewer egos than the largest shard sends a zero-padded block; 2 * HOLD of them, in rotation
            self.send = [like.new_zeros((self.rows, sh.n_pad)) for _ in range(2 * self.HOLD)] if sh.n_local != sh.n_pad else None
            # the zero fill must have RUN before a peer can learn this buffer's handle: its first push could
            # otherwise land before the fill and be wiped by it
            torch.cuda.synchronize(like.device)
        except Exception as e:                          # noqa: BLE001
            why = f"rank {sh.rank}: allocating / exporting the slot buffer failed: {e!r}"
        # every rank learns every rank's handle (None from a rank that failed); its own buffer is used through the
        # local pointer
        blobs = [None] * sh.world
        if sh.world > 1:
            dist.all_gather_object(blobs, bytes(ipc.bytes) if why is None else None, group=sh.group)
        ptrs = (C.c_void_p * sh.world)()
        if why is None and any(bl is None for r, bl in enumerate(blobs) if r != sh.rank):
            why = "a peer could not export its slot buffer"
        if why is None:
            try:
                for r in range(sh.world):
                    if r == sh.rank:
                        ptrs[r] = self._own
                        continue
                    peer, hd = C.c_void_p(), _lib.VdynIpcHandle()
                    C.memmove(hd.bytes, blobs[r], 64)
                    handle.call("vdyn_xchg_open", C.byref(hd), C.byref(peer))
                    self._peers.append(peer.value)
                    ptrs[r] = peer.value
            except Exception as e:                      # noqa: BLE001
                why = f"rank {sh.rank}: opening a peer's slot buffer failed: {e!r}"
        # destinations of a push: every rank's buffer but this rank's own (its block is placed by result())
        others = [ptrs[r] for r in range(sh.world) if r != sh.rank]
        self._dst, self._n_dst = (C.c_void_p * max(len(others), 1))(*others), len(others)
        if not self._agree(why is None):
            self.close()
            return None, why or "another rank could not set up its peer buffers"
        return self, None

    def start(self, term_local: torch.Tensor):
        assert tuple(term_local.shape) == (self.rows, self.sh.n_local)
        stream = self._C.c_void_p(torch.cuda.current_stream(term_local.device).cuda_stream)
        if len(self._inflight) >= 2 * self.HOLD:
            # the caller's stream waits, on the device, for every copy queued so far: all held sources have then been
            # read, and the HOLD oldest may be overwritten (the padded send blocks below) or freed (the allocator
            # reuses memory in stream order, and this wait is ahead of anything enqueued from here on)
            self.h.call("vdyn_xchg_fence", stream)
            del self._inflight[:self.HOLD]
        src = term_local.contiguous()
        if self.send is not None:
            src = self.send[self._count % (2 * self.HOLD)]
            src[:, :self.sh.n_local].copy_(term_local)
        self.h.call("vdyn_xchg_push", self._dst, self._n_dst, self.offsets[self.sh.rank],
                    self._C.c_void_p(src.data_ptr()), self.block, stream)
        self._inflight.append(src)              # stays referenced until the copies have read it
        self._last = src
        self._count += 1

    def wait(self):
        if self._inflight:
            try:
                self.h.call("vdyn_xchg_wait")
            finally:
                self._inflight = []

    def result(self) -> torch.Tensor:
        """[rows][n_units] of the last exchange, a private copy.  Collective: every rank must call it (two barriers).
        On return no rank reads a slot any more, so any rank may ``start`` the next exchange at once.  A rank whose
        own part fails still passes both barriers before it raises: its peers are never left inside one."""
        multi = dist.is_initialized() and self.sh.world > 1
        err, out = None, None
        try:
            self.wait()
        except Exception as e:                          # noqa: BLE001
            err = e
        if multi:
            dist.barrier(group=self.sh.group)   # every rank has waited for its own pushes
        try:
            if err is None:
                torch.cuda.synchronize(self.recv.device)
                if self._last is not None:      # this rank's own block: never sent anywhere, placed here
                    own = self.recv.view(self.sh.world, self.rows, self.sh.n_pad)[self.sh.rank]
                    own[:, :self._last.shape[1]].copy_(self._last)
                out = self.sh.assemble(self.recv, self.rows)
                torch.cuda.synchronize(self.recv.device)   # the copies out of the slots have run ...
        except Exception as e:                          # noqa: BLE001
            err = e
        if multi:
            dist.barrier(group=self.sh.group)   # ... on every rank, before a faster peer's next push overwrites one
        if err is not None:
            raise err
        return out

    def close(self):
        """Collective (two barriers: before the mappings go, and before the buffers themselves do): EVERY rank passes
        them, also one that has nothing to free -- the rank whose allocation failed in an agreed-failure path must not
        skip what its peers enter."""
        if self._closed:
            return
        self._closed = True
        err = None
        try:
            self.wait()
        except Exception as e:                          # noqa: BLE001
            err = e
        if dist.is_initialized() and self.sh.world > 1:
            dist.barrier(group=self.sh.group)   # nobody is still writing into a buffer about to go
        try:
            for p in self._peers:
                self.h.call("vdyn_xchg_close", self._C.c_void_p(p))
        except Exception as e:                          # noqa: BLE001
            err = err or e
        self._peers = []
        self.recv = None
        if dist.is_initialized() and self.sh.world > 1:
            dist.barrier(group=self.sh.group)   # every rank has unmapped its peers' buffers before any owner frees one
        if self._own is not None:
            self.h.call("vdyn_xchg_free", self._C.c_void_p(self._own))
        self._own = None
        if err is not None:
            raise err


def make_exchange(kind: str, sh: ShardedRollout, rows: int, like: torch.Tensor, handle=None):
    """`auto`: the peer-copy exchange when every rank can set it up AND one real exchange of a known pattern
    arrives intact on every rank (PeerExchange.try_create), the RCCL all-gather otherwise; the returned object says
    which (`kind`) and, after a fallback, why (`fallback_reason`)."""
    if kind == "auto":
        x, why = (None, "no library handle") if handle is None else PeerExchange.try_create(sh, rows, like, handle)
        if x is None:
            x = AllGatherExchange(sh, rows, like)
            x.fallback_reason = why
        return x
    if kind == "rccl":
        return AllGatherExchange(sh, rows, like)
    if kind == "p2p":
        if handle is None:
            raise ValueError("the peer-copy exchange needs the rank's library handle")
        return PeerExchange(sh, rows, like, handle)
    raise ValueError(f"unknown exchange {kind!r}")


def all_gather_argmin(best_cost_local: torch.Tensor, best_idx_local: torch.Tensor, n_ego, group=None):
    """MPC (config 5): egos are sharded, candidates are replicated, so the per-ego
    argmin is rank-local; gather the (cost, idx) pairs, 8 B per ego."""
    sh = ShardedRollout(n_ego, group, units_per_ego=1)
    c = sh.all_gather_terminal(best_cost_local[None, :])[0]
    i = sh.all_gather_terminal(best_idx_local[None, :])[0]
    return c, i
