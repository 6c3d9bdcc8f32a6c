#!/usr/bin/env python3
"""One secondary configuration of bench.py's `extra` block, launched a fixed number of times --
the program profiles/collect_kernels.sh puts after `rocprofv3 ... --` to get kernel-trace and PMC
evidence for the kernels that are not the headline.

    python3 profiles/kernels_workload.py --case NAME [--launches 20] [--manifest FILE]

--manifest writes what summarize.py needs to normalise the counters: the kernel's name, RK4 steps
per lane and launch, vehicle-steps per launch, algorithmic HBM bytes per launch
and the dynamic LDS bytes the launcher asks for (rocprofv3's LDS_Block_Size column only shows the
static part).  Kernel names are PREFIXES of the demangled name (trailing template flags such as TRAJ /
LOG select the instance the case happens to launch)."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CASES = ("headline", "per_rollout_controls", "config2_f64_lane", "config2_f64_wheel", "config2_f64_axles",
         "config2_f64_four_c", "config5_mpc",
         "closed_loop", "closed_loop_datalog", "trajectory_dump", "spiral_lattice")


def build(case, pkg, torch, dev):
    """-> (launch callable, manifest dict)"""
    W = pkg.workloads
    VM = pkg.VehicleModel
    vm = VM(2.906, np.deg2rad(30), 1e-3, device=0)
    if case in ("headline", "trajectory_dump", "per_rollout_controls"):
        n, H = 65536, 200
        s0, tab, pid = W.config3(n, H, np.float32)
        s0d, tabd, pidd = (torch.from_numpy(a).to(dev) for a in (s0, tab, pid))
        if case == "headline":
            return (lambda: vm.rollout(s0d, tabd, path_id=pidd)), dict(
                kernel="rollout_kernel<float, 2, 1, false, true", steps_per_lane=H, vehicle_steps=n * H,
                algo_bytes=96 * n + tab.nbytes + 4 * n, dynamic_lds_bytes=min(H, 48 * 1024 // (7 * 4 * 4)) * 7 * 4 * 4)
        if case == "trajectory_dump":
            return (lambda: vm.rollout(s0d, tabd, path_id=pidd, traj_stride=1)), dict(
                kernel="rollout_kernel<float, 2, 1, false, true", steps_per_lane=H, vehicle_steps=n * H,
                algo_bytes=96 * n + tab.nbytes + 4 * n + 48 * n * H,
                dynamic_lds_bytes=min(H, 48 * 1024 // (7 * 4 * 4)) * 7 * 4 * 4)
        ctrl = torch.from_numpy(W.expand_shared_controls(tab, pid)).to(dev)
        return (lambda: vm.rollout(s0d, ctrl)), dict(
            kernel="rollout_kernel<float, 2, 0, false, true", steps_per_lane=H, vehicle_steps=n * H,
            algo_bytes=96 * n + 8 * n * H, dynamic_lds_bytes=0)
    if case in ("config2_f64_lane", "config2_f64_wheel", "config2_f64_axles", "config2_f64_four_c"):
        H = 200
        s2, c2 = W.config2(64, H)
        s2d, c2d = torch.from_numpy(s2).to(dev), torch.from_numpy(c2).to(dev)
        if case.endswith("wheel"):
            vm = VM(2.906, np.deg2rad(30), 1e-3, device=0, lanes_per_rollout=4)
        if case.endswith(("axles", "four_c")):
            # tires that differ by axle (two fits pinned in registers, rollout_kernel<..., PW = 2>) or by wheel
            # (the per-wheel table in LDS, PW = 1): csrc/vdyn_kernels.hip fit_mode
            tires = pkg.VehicleParameters()
            tires.CRL = tires.CRR = 1.3
            if case.endswith("four_c"):
                tires.CFR, tires.CRR = 1.45, 1.25
            vm = VM(2.906, np.deg2rad(30), 1e-3, device=0, params=tires)
        return (lambda: vm.rollout(s2d, c2d)), dict(
            kernel=("rollout_quad_kernel<double, 2, 0, true, false" if case.endswith("wheel")
                    else "rollout_kernel<double, 2, 0, false, true, false, 2" if case.endswith("axles")
                    else "rollout_kernel<double, 2, 0, false, true, false, 1" if case.endswith("four_c")
                    else "rollout_kernel<double, 2, 0, false, true, false, 0"),
            steps_per_lane=H, vehicle_steps=4096 * H, algo_bytes=192 * 4096 + 16 * 4096 * H, dynamic_lds_bytes=0)
    if case == "config5_mpc":
        E, C, H = 1024, 512, 50
        ego, cand, goal = (torch.from_numpy(a).to(dev) for a in W.config5(E, C, H))
        return (lambda: vm.mpc_argmin(ego, cand, goal, dt=2e-3, w_delta=W.MPC_W_DELTA)), dict(
            # egos on the lanes: a wave = 64 egos x a chunk of KC = 4 candidates (launch_mpc_argmin), 4 x 50 steps per lane
            kernel="mpc_argmin_lanes_kernel<float, true", steps_per_lane=4 * H, vehicle_steps=E * C * H,
            algo_bytes=(12 + 2 + 2) * 4 * E + cand.numel() * 4, dynamic_lds_bytes=0)
    if case in ("closed_loop", "closed_loop_datalog"):
        n = 65536
        H = 100 if case.endswith("datalog") else 200
        cl = [torch.from_numpy(a).to(dev) for a in W.closed_loop_config(n, dtype=np.float32)]
        dl = case.endswith("datalog")
        Wp, P = 1024, 7
        # ClosedLoopLds (csrc/vdyn_controls.hpp): x / y rows, cumulative arcs, circle rows per 32 and per 8 waypoints
        nb, nsb = (Wp + 31) // 32, (Wp + 7) // 8
        ws, segs = nsb * 8 + 4, (Wp + 3) // 4 * 4 + 4
        brs, srs = (nb + 15) // 16 * 16 + 4, (nsb + 15) // 16 * 16 + 20
        lds = (2 * P * ws + P * segs + 3 * P * (brs + srs)) * 4
        return (lambda: vm.closed_loop(cl[0], cl[1], cl[2], H, wcount=cl[3], path_id=cl[4], datalog=dl)), dict(
            kernel=f"closed_loop_kernel<float, true, true, {'true' if dl else 'false'}", steps_per_lane=H,
            vehicle_steps=n * H, algo_bytes=(24 + 12) * 4 * n + 4 * n + cl[2].numel() * 4 + (180 * n * H if dl else 0),
            dynamic_lds_bytes=lds)
    if case == "spiral_lattice":
        n, H = 65536, 200
        s0, sp = W.config3_spiral(n, H, np.float32)
        s0d, spd = torch.from_numpy(s0).to(dev), torch.from_numpy(sp).to(dev)
        return (lambda: vm.rollout_spiral(s0d, spd, H)), dict(
            kernel="rollout_spiral_kernel<float", steps_per_lane=H, vehicle_steps=n * H,
            algo_bytes=96 * n + sp.nbytes, dynamic_lds_bytes=0)
    raise SystemExit(f"unknown case {case}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", required=True, choices=CASES)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--warm-ms", type=float, default=60.0)
    ap.add_argument("--manifest", default=None)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module("python-motionplanning_amd")
    dev = torch.device("cuda:0")
    fn, man = build(args.case, pkg, torch, dev)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    warm = 0
    while (time.perf_counter() - t0) * 1e3 < args.warm_ms:
        fn()
        torch.cuda.synchronize()
        warm += 1
    ev = []
    for _ in range(args.launches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    d = np.array([a.elapsed_time(b) for a, b in ev])
    man.update(case=args.case, launches=args.launches, warm_launches=warm + 1,
               event_ms_mean=float(d.mean()), event_ms_median=float(np.median(d)))
    if args.manifest:
        with open(args.manifest, "w") as f:
            json.dump(man, f, indent=1)
    print(json.dumps(man))


if __name__ == "__main__":
    main()
