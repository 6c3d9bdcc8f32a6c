import sys, importlib, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("python-motionplanning_amd")
W = pkg.workloads
dev = torch.device("cuda:0")
vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-3, device=0)
s0, c0, wp, wc, pid = W.closed_loop_config(65536, dtype=np.float32)
t = lambda a: torch.from_numpy(a).to(dev)
s0, c0, wp, wc, pid = t(s0), t(c0), t(wp), t(wc), t(pid)
def run(tag, **kw):
    g = pkg._lib.default_ctrl_gains()
    for k, v in kw.pop("gains", {}).items(): setattr(g, k, v)
    f = lambda: vm.closed_loop(s0, c0, wp, 200, wcount=wc, path_id=pid, gains=g, **kw)
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.06:
        f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print(f"{tag:40s} {a.elapsed_time(b)/10:.3f} ms")
run("default (20 updates)")
run("lookahead = 0 (no walk)", gains={"lookahead": 0.0})
run("ctrl_every = 100 (2 updates)", ctrl_every=100)
run("ctrl_every = 1000 (1 update)", ctrl_every=1000)
run("ctrl_every = 5 (40 updates)", ctrl_every=5)
