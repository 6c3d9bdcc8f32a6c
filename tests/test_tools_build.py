"""CPU: the stand-alone kernel harnesses under tools/ubench/ still compile against the current sources (hipcc
cross-compiles gfx950 without a GPU).  They include csrc/vdyn_kernels.hip directly and call launchers and kernels by
name, so a signature change in the library would otherwise rot them silently.  Syntax and template instantiation only
(-fsyntax-only, device side): seconds per file, nothing is generated or run."""
import os
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-fno-fast-math", "--cuda-device-only", "-fsyntax-only",
         "-I", os.path.join(REPO, "python-motionplanning_amd", "csrc")]


@pytest.mark.parametrize("src,defs", [("mpc_harness.hip", []), ("f64_harness.hip", []), ("cl_harness.hip", []),
                                      ("cl_harness.hip", ["-DVDYN_STAMPS"]), ("store_rate.hip", []),
                                      ("rcp64_probe.hip", [])])
def test_ubench_harness_compiles(src, defs):
    import importlib
    hipcc = importlib.import_module("python-motionplanning_amd._build").hipcc_path()
    res = subprocess.run([hipcc, *FLAGS, *defs, os.path.join(REPO, "tools", "ubench", src)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
