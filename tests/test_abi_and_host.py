"""CPU: the C-ABI library loads and exports every symbol include/vdyn.h declares
(no compute calls: there is no GPU here), the ctypes binding covers exactly that
set, and the host-side mirror behaves like the reference interface."""
import ctypes
import importlib
import os
import re
import subprocess

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "vdyn.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vdyn_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib(pkg):
    bld = importlib.import_module("python-motionplanning_amd._build")
    bld.build()          # hipcc cross-compiles gfx950 without a GPU
    return importlib.import_module("python-motionplanning_amd._lib")


def test_header_symbols_all_exported(lib):
    syms = declared_symbols()
    assert len(syms) == 13 + 2 * 26 + 7, syms          # + the seven vdyn_xchg_* entry points; 13 incl. vdyn_build_id
    dll = ctypes.CDLL(lib.LIB_PATH)
    for s in syms:
        assert hasattr(dll, s), f"{s} declared in include/vdyn.h but not exported"
    assert sorted(lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    nm = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r" T (vdyn_\w+)", nm)))
    assert exported == syms, "library exports exactly the declared C symbols"


def test_library_carries_gfx950_code_only(lib):
    out = subprocess.run(["strings", "-a", lib.LIB_PATH], capture_output=True, text=True).stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx\w+)", out))
    assert archs == {"gfx950"}, archs


def test_params_struct_and_defaults(lib):
    assert ctypes.sizeof(lib.VdynParams) == 19 * 8 and ctypes.sizeof(lib.VdynCtrlGains) == 9 * 8
    assert lib.load().vdyn_abi_version() == lib.VDYN_ABI_VERSION
    bld = importlib.import_module("python-motionplanning_amd._build")
    assert lib.build_id() == bld.source_hash(), "the library carries the hash of the sources it was built from"
    p = lib.default_params()
    # SURVEY.md section 8(a1), measured on the reference's VehicleParameters()
    assert p.m == pytest.approx(1857.82, abs=1e-12)
    assert p.b == 1.5708108108108108 and p.a == 1.3351891891891894
    assert p.Izz == 1948.2304506781593
    assert p.rw == 0.308309813617345
    assert list(p.B) == [20.6357] * 4 and list(p.C) == [1.5047] * 4 and p.g == 9.81


def test_vehicle_parameters_mirror(pkg, lib):
    P = pkg.VehicleParameters()
    c = pkg.vehicle_model.params_to_c(P)
    assert c.key() == lib.default_params().key()
    for name in ("rr", "mus", "mf", "mr", "m", "L", "ab_ratio", "b", "a", "Izz", "Jw", "hg", "T", "kf", "kr",
                 "rw", "BFL", "CFL", "DFL", "BFR", "CFR", "DFR", "BRL", "CRL", "DRL", "BRR", "CRR", "DRR",
                 "Efront", "Erear", "E", "LeverArm", "wL", "wR"):
        assert hasattr(P, name), name       # attribute set of vehicle_model.py:23-61
    assert P.DFL == 1.1233 and P.E == [0.0376, 0.0376, 0, 0]
    Q = pkg.VehicleParameters(mf=1000.0, mr=800.0, L=3.0, ab_ratio=1.0, BFL=10.0)
    assert Q.m == 1800.0 and Q.a == Q.b == 1.5 and Q.BRR == 10.0


def test_no_gpu_means_loud_failure(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    vm = pkg.VehicleModel(2.906, np.deg2rad(30), 1e-4)       # loads the library: fine without a GPU
    with pytest.raises(pkg.VdynError) as e:
        vm.planar_model_RK4([25.0, 0, 0, 81.0, 81.0, 81.0, 81.0, 0, 0, 0], [0] * 4, [1.0] * 4,
                            [0.02, 0.02, 0, 0], pkg.VehicleParameters(), 0, 0)
    assert e.value.code == -3 and "no HIP device" in str(e.value)
    with pytest.raises(pkg.VdynError):
        vm.rollout(np.zeros((12, 4)), np.zeros((3, 2, 4)))   # no CPU fallback, by design


def test_argument_validation_precedes_device_use(pkg):
    vm = pkg.VehicleModel(1.0, 0.7, 1e-3)
    P = pkg.VehicleParameters()
    with pytest.raises(ValueError):
        vm.planar_model_RK4([1.0] * 9, [0] * 4, [1] * 4, [0] * 4, P, 0, 0)
    with pytest.raises(ValueError):
        vm.planar_model([1.0] * 10, [0] * 3, [1] * 4, [0] * 4, P, 0, 0)
    with pytest.raises(ValueError):
        vm.step(np.zeros((10, 3)), np.zeros((2, 3)))
    with pytest.raises(ValueError):
        vm.step(np.zeros((12, 3)), np.zeros((5, 3)))
    with pytest.raises(ValueError):
        vm.rollout(np.zeros((12, 3)), np.zeros((4, 2, 5)))
    with pytest.raises(ValueError):
        vm.rollout(np.zeros((12, 3)), np.zeros((2, 4, 2)), path_id=[0, 1, 2])
    with pytest.raises(ValueError):
        vm.rollout(np.zeros((12, 3)), np.zeros((4, 2, 3)), mu_max=[1, 1, 1])
    with pytest.raises(ValueError):
        vm.mpc_argmin(np.zeros((12, 3)), np.zeros((4, 3, 8)), np.zeros((2, 3)))
    with pytest.raises(ValueError):
        vm.rollout(np.zeros((12, 3), dtype=np.float64), np.zeros((4, 2, 3)), traj_stride=-1)


def test_product_never_touches_the_oracle():
    """The package must not import, link or load anything under oracle/."""
    pkg_dir = os.path.join(REPO, "python-motionplanning_amd")
    for root, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(root, f), errors="replace").read()
                code = "\n".join(l for l in txt.split("\n") if not l.strip().startswith(("#", "//", "*", '"')))
                assert not re.search(r"^\s*(from|import)\s+oracle|libvdyn_oracle|vdyn_oracle\.h|import_module\(.oracle",
                                     code, flags=re.M), \
                    f"{f} references the oracle"


def test_workloads_are_deterministic_and_shaped(workloads):
    s, c = workloads.config2(64, 200)
    assert s.shape == (12, 4096) and c.shape == (200, 2, 4096) and s.dtype == np.float64
    assert c[0, 0, 0] == -0.3 and c[0, 0, -1] == 0.3 and c[0, 1, 0] == -200.0 and c[0, 1, 63] == 400.0
    a = workloads.config3(65536, 200)
    b = workloads.config3(65536, 200)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    s0, tab, pid = a
    assert s0.shape == (12, 65536) and s0.dtype == np.float32 and tab.shape == (7, 200, 2)
    assert pid.dtype == np.int32 and pid.max() == 6
    assert np.array_equal(s0[:, 0], s0[:, 6]) and not np.array_equal(s0[:, 6], s0[:, 7])   # ego = r // 7
    assert np.allclose(tab[3, :, 0], 0) and np.all(tab[:, :, 1] == 100.0)
    assert (s0[0] >= 10).all() and (s0[0] <= 30).all() and (s0[10:] == 0).all()
    e, cd, g = workloads.config5(1024, 512, 50)
    assert e.shape == (12, 1024) and cd.shape == (50, 2, 512) and g.shape == (2, 1024)
    assert np.array_equal(cd[0], cd[9]) and not np.array_equal(cd[9], cd[10])              # 10-step hold
    assert np.abs(cd[:, 0]).max() <= 0.5236 + 1e-7
    exp = workloads.expand_shared_controls(tab, pid[:21])
    assert exp.shape == (200, 2, 21) and np.array_equal(exp[:, :, 8], tab[1])


def test_register_counts_come_from_the_code_object(pkg):
    """profiles/summarize*.py report vgpr / agpr / sgpr / spills from the metadata of the gfx950 code objects inside the
    built library (tools/isa/code_object_meta.py), not from rocprofv3's VGPR_Count column (which says 108 for the
    headline kernel whose code object says 212: VERDICT round 3, weak 4).  No GPU needed: the library is only read."""
    import os
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools", "isa"))
    import code_object_meta as M
    meta = M.kernel_meta()
    assert len(meta) > 100, "two translation units' worth of kernels"
    head = M.lookup(meta, "void vdyn::rollout_kernel<float, 2, 1, false, true, false, 0, false>(vdyn::DevParams<float>, long)")
    assert head is not None and 150 <= head["vgpr"] <= 256 and head["agpr"] == 0 and head["scratch_bytes"] == 0
    assert head["sgpr"] <= 108 and head["max_flat_workgroup_size"] == 256
    # a name as rocprofv3 truncates it still finds its kernel; an ambiguous prefix finds none
    assert M.lookup(meta, "vdyn::rollout_kernel<float, 2, 1, false, true, false, 0, false>") == head
    assert M.lookup(meta, "vdyn::rollout_kernel<float") is None
