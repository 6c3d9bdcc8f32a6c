#!/usr/bin/env python3
"""The fp32 closed-loop seed sweep is a -m gpu test now (tests/test_gpu_controllers32.py::test_fp32_closed_loop_seed_sweep
and ..._pooled); this runs exactly those and leaves the table in gpurun_out/closed_loop_seed_sweep.txt.
usage (GPU box): python tools/closed_loop_seed_sweep.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "-q", "-s", "-m", "gpu", "-k", "seed_sweep",
                          os.path.join(ROOT, "tests", "test_gpu_controllers32.py")], cwd=ROOT))
