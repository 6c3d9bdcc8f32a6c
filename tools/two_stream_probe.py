#!/usr/bin/env python3
"""Does alternating the launches of independent batches between two HIP streams hide the
dispatch gap between back-to-back kernels?  Run on the GPU box: python tools/two_stream_probe.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("python-motionplanning_amd")
W = pkg.workloads
dev = torch.device("cuda:0")
vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3)
s0, tab, pid = (torch.from_numpy(a).to(dev) for a in W.config3(65536, 200, np.float32))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run(k, two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(k):
        if two:
            with torch.cuda.stream(streams[i & 1]):
                vm.rollout(s0, tab, path_id=pid)
        else:
            vm.rollout(s0, tab, path_id=pid)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3

run(1000, False)
for _ in range(3):
    print(f"one stream {run(400, False):.4f} ms/launch   two streams {run(400, True):.4f} ms/launch")
