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
    full = None
    for it in range(3):                                               # three steps: slots are rewritten in place
        term = vm.rollout(s0d, tabd, path_id=pidd)
        x.wait()
        x.start(term)
    full = x.result()
    np.save(os.path.join(out_dir, f"p2p_rank{rank}.npy"), full.cpu().numpy())
    x.close()
    dist.barrier()
    dist.destroy_process_group()
