"""What a handle pays when it cannot take the fitted tire chain: configs[1] (fp64, 4096 x 200)
and configs[2] (fp32, 65536 x 200) with the reference's tires, with a different C on the rear axle (fp32: four
fits; fp64: one coefficient set per handle, so the general atan -> sine chain) and with C = 2.3 (fitted since the
condition became "B >= 0 and the fit validates"), lane-per-rollout and wheel-parallel kernels.

    python3 tools/tire_chain_probe.py          # on the GPU box

Measured (round 2): default 0.43 / 0.157 ms; rear C = 1.3: fp64 0.87 ms (general chain, 2 x), fp32 0.157 ms;
C = 2.3: 0.43 / 0.156 ms.  Wheel-parallel fp64: 0.24 vs 0.39 ms."""
import sys, importlib, numpy as np, torch, time
sys.path.insert(0, '.')
pkg = importlib.import_module("python-motionplanning_amd")
W = pkg.workloads
dev = torch.device("cuda:0")
s2, c2 = W.config2(64, 200)
s2d, c2d = torch.from_numpy(s2).to(dev), torch.from_numpy(c2).to(dev)
s3, tab, pid = W.config3(65536, 200, np.float32)
s3d, tabd, pidd = (torch.from_numpy(a).to(dev) for a in (s3, tab, pid))
def timeit(f, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3: f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
VP = pkg.VehicleParameters
pw = VP(); pw.CRL = pw.CRR = 1.3
big = VP(CFL=2.3)
for name, veh in (("default", VP()), ("per-wheel C", pw), ("C = 2.3", big)):
    for lanes in (1, 4):
        vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, params=veh, device=0, lanes_per_rollout=lanes)
        print(f"{name:12s} lanes {lanes}: fp64 4096x200 {timeit(lambda: vm.rollout(s2d, c2d)):.4f} ms   fp32 65536x200 {timeit(lambda: vm.rollout(s3d, tabd, path_id=pidd)):.4f} ms")
