"""Diagnostic (GPU box): what the fp32 closed loop does on the adversarial NaN-waypoint table (table 4 of
tests/test_gpu_controllers32.py::_adversarial_tables) -- which vehicles turn non-finite, when, and with which log rows.
usage: python tests/diag_nan_table.py"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("python-motionplanning_amd")
T = importlib.import_module("test_gpu_controllers32")

wp, wc = T._adversarial_tables()
rng = np.random.default_rng(99)
n, dt, H = 1536, 1e-3, 40
pid = (np.arange(n) % 6).astype(np.int32)
k = (rng.uniform(0, 1, n) * (wc[pid] - 1)).astype(int)
k[pid == 4] = np.maximum(k[pid == 4], 10)
s0 = np.zeros((12, n))
s0[0] = rng.uniform(8, 25, n)
s0[3:7] = s0[0] / 0.308309813617345
far = rng.uniform(0, 1, n) < 0.05
s0[8] = np.nan_to_num(wp[pid, k, 0]) + rng.normal(0, 0.3, n) + 200.0 * far
s0[9] = np.nan_to_num(wp[pid, k, 1]) + rng.normal(0, 0.3, n)
s0[8, pid == 0] *= rng.uniform(0, 1, (pid == 0).sum()) < 0.5
s0[9, pid == 0] *= s0[8, pid == 0] != 0
s0[7] = rng.uniform(-np.pi, np.pi, n)
c0 = np.zeros((6, n))
c0[2], c0[3] = s0[0], 25.0
vm = pkg.VehicleModel(2.906, np.deg2rad(30), dt, device=0)
f = lambda a: a.astype(np.float32)
t32, c32, log = vm.closed_loop(f(s0), f(c0), f(wp), H, wcount=wc, path_id=pid, log=True)
t4 = np.where(pid == 4)[0]
bad = [i for i in t4 if not np.isfinite(log[:, :, i]).all()]
print(f"table 4: {len(t4)} vehicles, {len(bad)} with a non-finite log entry; far among them: {int(far[bad].sum())}")
for i in bad[:12]:
    nf = ~np.isfinite(log[:, :, i])
    t_first = int(np.where(nf.any(axis=1))[0][0])
    print(f" vehicle {i}: k0 {k[i]} far {bool(far[i])} x0 {s0[8, i]:.3f} y0 {s0[9, i]:.3f} yaw0 {s0[7, i]:.3f}; first bad step {t_first}, "
          f"bad rows there {np.where(nf[t_first])[0].tolist()}; idx/cte at updates: "
          + " ".join(f"{int(log[t, 14, i]) if np.isfinite(log[t, 14, i]) else 'nan'}/{log[t, 15, i]:.3g}" for t in range(0, H, 10)))
# one controller update alone, same inputs: raw steering, index, crosstrack error
cs, out = vm.controller_update(f(s0), f(c0), f(wp), wcount=wc, path_id=pid)
b2 = [i for i in t4 if not np.isfinite(out[:, i]).all()]
print(f"controller_update alone (global-memory tables): {len(b2)} non-finite of table 4; e.g. {[(int(i), out[:, i].tolist()) for i in b2[:5]]}")
