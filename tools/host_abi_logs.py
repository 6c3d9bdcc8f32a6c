"""GPU box: the two host-pointer calls whose OUTPUT is large, through the raw C ABI with caller-owned, already-touched
output arrays (what the library itself spends; a fresh np.empty of 1.2 GB adds its page faults on top):
  * vdyn_closed_loop_f32_host with the 45-column DataLog, 65536 vehicles x 100 sub-steps: 1.18 GB out;
  * vdyn_rollout_f32_host with per-rollout controls and every state written (traj_stride 1), 65536 x 200: 105 MB in, 629 MB out;
  * the same with the controls as the lattice's shared table [7][200][2]: 629 MB out.
usage: python tools/host_abi_logs.py            (VDYN_LIB_PATH selects another library build for an A/B)"""
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("python-motionplanning_amd")
L = pkg._lib
W = pkg.workloads


def vp(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, min(ts) * 1e3


if __name__ == "__main__":
    vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, device=0)
    h = vm.handle()
    out = {"build_id": L.build_id()}
    n, H = 65536, 100
    st, cs, wp, wc, pid = W.closed_loop_config(n, dtype=np.float32)
    g = L.default_ctrl_gains()
    term, cso = np.zeros((12, n), np.float32), np.zeros((6, n), np.float32)
    dl = np.zeros((H, 45, n), np.float32)
    call = lambda: h.call("vdyn_closed_loop_f32_host", C.byref(g), n, H, 10, 0, vp(st), vp(cs), vp(wp), wp.shape[1], vp(wc), vp(pid),
                          wp.shape[0], 1e-3, vp(term), vp(cso), None, vp(dl))
    med, best = timed(call)
    out["closed_loop_datalog_65536x100_f32_host"] = {"ms": med, "ms_min": best, "GB_out": dl.nbytes / 1e9, "GBs": dl.nbytes / med / 1e6}
    del dl
    s0, tab, pidr = W.config3(n, 200, np.float32)
    ctrl = W.expand_shared_controls(tab, pidr)
    term = np.zeros((12, n), np.float32)
    traj = np.zeros((200, 12, n), np.float32)
    call = lambda: h.call("vdyn_rollout_f32_host", n, 200, vp(s0), vp(ctrl), 2, 0, None, 0, 1e-3, None, vp(term), vp(traj), 1)
    med, best = timed(call)
    out["rollout_per_rollout_controls_traj1_65536x200_f32_host"] = {"ms": med, "ms_min": best, "GB_in": ctrl.nbytes / 1e9,
                                                                     "GB_out": traj.nbytes / 1e9}
    call = lambda: h.call("vdyn_rollout_f32_host", n, 200, vp(s0), vp(tab), 2, 1, vp(pidr), tab.shape[0], 1e-3, None, vp(term), vp(traj), 1)
    med, best = timed(call)
    out["rollout_shared_table_traj1_65536x200_f32_host"] = {"ms": med, "ms_min": best, "GB_out": traj.nbytes / 1e9}
    import torch
    pin = torch.empty(traj.nbytes // 4, dtype=torch.float32).pin_memory()
    d = torch.empty(traj.nbytes // 4, dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
    for a_, b_ in ev:
        a_.record()
        pin.copy_(d, non_blocking=True)
        b_.record()
    torch.cuda.synchronize()
    t = float(np.median([a_.elapsed_time(b_) for a_, b_ in ev]))
    out["pinned_d2h"] = {"ms_for_629_MB": t, "GBs": traj.nbytes / t / 1e6}
    print(json.dumps(out))
