import importlib, numpy as np, torch, sys
sys.path.insert(0, '.')
pkg = importlib.import_module("python-motionplanning_amd")
W = pkg.workloads
dev = torch.device("cuda:0")
vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, device=0)
def timed(fn, reps=30):
    for _ in range(200): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for n in (65536, 32768, 16384, 8192, 4096):
    s0, tab, pid = W.config3(n, 200, np.float32)
    s0d, tabd, pidd = (torch.from_numpy(a).to(dev) for a in (s0, tab, pid))
    res = [n]
    for stride in (0, 1, 2, 4, 25):
        res.append(round(timed(lambda: vm.rollout(s0d, tabd, path_id=pidd, traj_stride=stride)), 4))
    print("n, ms at traj_stride 0/1/2/4/25:", res, flush=True)
