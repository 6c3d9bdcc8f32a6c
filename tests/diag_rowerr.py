"""Diagnostic (run by hand on a GPU box: `python tests/diag_rowerr.py`), not collected by pytest: per-row error of the
fp32 HIP rollout and of the plain-C FLOAT oracle against the fp64 oracle on BASELINE configs[2] -- the evidence that
the fp32 error is the reference's own fp32 floor.  Lives under tests/ because it uses oracle/."""
import sys, importlib, numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module("python-motionplanning_amd")
from oracle import oracle as O
W = pkg.workloads
vm = pkg.VehicleModel(2.906, 0.52, 1e-3)
s0, tab, pid = W.config3(65536, 200)
t32 = vm.rollout(s0, tab, path_id=pid)
want = O.rollout(O.default_params(), s0.astype(np.float64), tab.astype(np.float64), 1e-3, path_id=pid, nthreads=64)
o32 = O.rollout(O.default_params(), s0, tab, 1e-3, path_id=pid, nthreads=64)
for i, nme in enumerate(W.ROW_NAMES):
    e = np.abs(t32[i] - want[i]); eo = np.abs(o32[i].astype(np.float64) - want[i])
    print(f"{nme:8s} scale {np.abs(want[i]).max():9.3f}  gpu32 maxabs {e.max():.2e} rms {np.sqrt((e**2).mean()):.2e} | cpu-float-oracle maxabs {eo.max():.2e} rms {np.sqrt((eo**2).mean()):.2e}")
