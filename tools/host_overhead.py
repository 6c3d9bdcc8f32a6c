"""Host-side cost of one rollout launch through the Python shim (no GPU wait inside the loop)."""
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("python-motionplanning_amd")
W = pkg.workloads
dev = torch.device("cuda:0")
vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, device=0)
s0, tab, pid = (torch.from_numpy(a).to(dev) for a in W.config3(65536, 200, np.float32))


def loop(k, events):
    t0 = time.perf_counter()
    for _ in range(k):
        if events:
            a = torch.cuda.Event(enable_timing=True)
            a.record()
        vm.rollout(s0, tab, path_id=pid)
        if events:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / k * 1e3, (time.perf_counter() - t0) / k * 1e3


for _ in range(3):
    loop(100, False)
for k in (50, 200, 800):
    for ev in (False, True):
        print(f"k={k} events={ev}: host enqueue {loop(k, ev)[0]:.4f} ms/launch, wall {loop(k, ev)[1]:.4f}")
pr = cProfile.Profile()
pr.enable()
loop(200, True)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


def loop_retained(k, precreate):
    keep = []
    pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(k)] if precreate else None
    t0 = time.perf_counter()
    for i in range(k):
        a, b = pool[i] if precreate else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        a.record()
        term = vm.rollout(s0, tab, path_id=pid)
        b.record()
        keep.append((a, b))
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = np.array([a.elapsed_time(b) for a, b in keep])
    return (t1 - t0) / k * 1e3, (t2 - t0) / k * 1e3, d.mean(), np.median(d)


for rep in range(2):
    for k in (20, 200, 400):
        for pre in (False, True):
            print(f"retained k={k} precreate={pre}: enqueue %.4f wall %.4f kernel mean %.4f median %.4f" % loop_retained(k, pre))
