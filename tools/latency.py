#!/usr/bin/env python3
"""Single-vehicle call latency (BASELINE configs[0] pattern): the reference-signature drop-in,
and the bare C ABI call underneath it.  Run on the GPU box: python tools/latency.py"""
import ctypes as C
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("python-motionplanning_amd")

vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-4)
pp = vm.params
st, axp, ayp = [25.0, 0, 0] + [25.0 / pp.rw] * 4 + [0, 0, 0], 0.0, 0.0
for _ in range(50):
    o = vm.planar_model_RK4(st, [50.0] * 4, [1.0] * 4, [0.02, 0.02, 0, 0], pp, axp, ayp)
n = 2000
t0 = time.perf_counter()
for _ in range(n):
    o = vm.planar_model_RK4(st, [50.0] * 4, [1.0] * 4, [0.02, 0.02, 0, 0], pp, axp, ayp)
    st, axp, ayp = o[0], o[7], o[8]
t_drop = (time.perf_counter() - t0) / n
h, a = vm._scalar_handle(pp), vm._sc_addr
t0 = time.perf_counter()
for _ in range(n):
    h.call("vdyn_step_f64_host", 1, a[0], a[24], 12, 1e-4, None, a[40], a[52], a[62])
t_abi = (time.perf_counter() - t0) / n
print(f"drop-in planar_model_RK4: {t_drop * 1e6:.1f} us/call   bare vdyn_step_f64_host: {t_abi * 1e6:.1f} us/call   "
      f"(reference NumPy: 247.7 us/call)")
