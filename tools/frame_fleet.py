#!/usr/bin/env python3
"""One whole Car.drive frame (drive.py:112-154) for a fleet of egos, device to device -- the
workload of bench.py's extra.full_frame_fleet_4096_f64, alone, for rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_frame -- python3 tools/frame_fleet.py [f32|f64]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    dt = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == "f32") else np.float64
    pkg = importlib.import_module("python-motionplanning_amd")
    W = pkg.workloads
    dev = torch.device("cuda", 0)
    Ef = 4096
    th = np.linspace(0.0, 2 * np.pi, 4000, endpoint=False)
    gpx, gpy = 200.0 * np.cos(th), 200.0 * np.sin(th)
    k = np.random.default_rng(20244).integers(0, 4000, Ef)
    ego = np.stack([gpx[k] + 0.5, gpy[k] - 0.5, th[k] + np.pi / 2 + 0.05])
    gx, gy, egof = (torch.from_numpy(a.astype(dt)).to(dev) for a in (gpx, gpy, ego))
    obst = torch.from_numpy(np.stack([gpx[::97] * 1.02, gpy[::97] * 1.02], axis=1).astype(dt)).to(dev)
    s_f = np.zeros((12, Ef), dt)
    s_f[0], s_f[3:7] = 25.0, 25.0 / W.DEFAULT_RW
    s_f[[8, 9, 7]] = ego
    c_f = np.zeros((6, Ef), dt)
    c_f[2], c_f[3] = 25.0, 25.0
    s_f, c_f = torch.from_numpy(s_f).to(dev), torch.from_numpy(c_f).to(dev)
    ids = torch.arange(Ef, dtype=torch.int32, device=dev)
    vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-4, device=0)

    def frame():
        lat = vm.plan_lattice(gx, gy, egof, 25.0)
        gi = lat["goal_index"].long()
        goal = torch.stack([gx[gi], gy[gi]])
        _, best, _ = vm.select_best_path(lat["paths"], obst, goal, validity=lat["validity"])
        wp, wc = vm.interpolate_waypoints(lat["paths"], best, 0.01, 4096)
        return vm.closed_loop(s_f, c_f, wp, 100, wcount=wc, path_id=ids)

    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        term, _ = frame()
    torch.cuda.synchronize()
    print(f"{np.dtype(dt).name}: {(time.perf_counter() - t0) / 3 * 1e3:.3f} ms per frame of {Ef} egos; finite:",
          bool(torch.isfinite(term).all()))


if __name__ == "__main__":
    main()
