"""Rank entry of tests/test_gpu_multishard.py::test_peer_copy_exchange_two_processes_one_gpu:
two processes share cuda:0, each integrates its whole-ego shard with the HIP kernel and the
blocks travel through distributed.PeerExchange (IPC handles + device-to-device copies)."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

if __name__ == "__main__":
    out_dir, n, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)      # side channel for the 64-byte handles
    pkg = importlib.import_module("python-motionplanning_amd")
    D = importlib.import_module("python-motionplanning_amd.distributed")
    W = pkg.workloads
    dev = torch.device("cuda:0")                                      # every rank on the one GPU of the box
    vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, device=0)
    s0, tab, pid = W.config3(n, H, np.float32)
    sh = D.ShardedRollout(n)
    s0d = torch.from_numpy(np.ascontiguousarray(s0[:, sh.lo:sh.hi])).to(dev)
    pidd = torch.from_numpy(pid[sh.lo:sh.hi].copy()).to(dev)
    tabd = torch.from_numpy(tab).to(dev)
    x = D.make_exchange("p2p", sh, 12, s0d, handle=vm.handle())
    # Twelve exchanges back to back, no host wait in between (the steady state of bench.py's step): eleven blocks of
    # distinct constants, then the real terminal states.  Pushes follow each other on the device, so the LAST one
    # wins in every slot of every rank; more than 2 * HOLD of them, so the device-side fence runs (at the ninth) and
    # releases the oldest sources, and a padded send block (ragged n) is reused after a full turn of its ring.
    term = vm.rollout(s0d, tabd, path_id=pidd)
    assert 11 > 2 * x.HOLD
    for it in range(11):
        x.start(torch.full_like(term, float(100 * (rank + 1) + it)))
    x.start(term)
    assert len(x._inflight) <= 2 * x.HOLD, "the fence must have released the oldest sources"
    full = x.result()
    # ... and once more after a result(): slots are rewritten in place, the host-side wait has reset the ring
    x.start(torch.full_like(term, -1.0))
    x.start(vm.rollout(s0d, tabd, path_id=pidd))
    again = x.result()
    assert torch.equal(again, full), "the second round of exchanges must deliver the same blocks"
    np.save(os.path.join(out_dir, f"p2p_rank{rank}.npy"), full.cpu().numpy())
    x.close()
    dist.barrier()
    dist.destroy_process_group()
