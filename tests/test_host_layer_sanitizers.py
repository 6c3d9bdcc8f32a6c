"""CPU: the library's OWN host layer under AddressSanitizer + UndefinedBehaviorSanitizer ("race
detection / sanitizers").  csrc/vdyn_capi.hip -- handle, argument checks, `Stage` offsets, scratch
growth, the pipelined upload of per-rollout controls, the peer-exchange calls -- is compiled as plain C++ against
tests/hipstub/ (a stand-in for the HIP runtime whose streams run their work LATE and in random order, and for the kernel
launchers, whose "kernels" touch every byte their arguments promise) and driven through every `_host` entry point by
tests/_host_layer_driver.py in a child interpreter with libasan preloaded.  A stand-in for a library inside the product's
own test: no oracle, no CPU path of the product.

What the deferred stub is worth: tests/hipstub/negative_experiments.sh removes, one at a time, six ordering dependencies
of the upload / download pipelines from a copy of vdyn_capi.hip (the host waiting for chunk c - 2's upload before
refilling its pinned buffer, the upload stream waiting for chunk c - 2's kernel, the compute stream waiting for chunk
c's upload, the download waiting for the kernel, the host waiting for the download, the closed loop's first chunk ending
at a controller period): the driver then fails under 4 of 4 seeds, every time (profiles/
r05_host_layer_negative_experiments.txt; two and a half minutes, so not part of this suite)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(REPO, "tests", "hipstub")


@pytest.fixture(scope="module")
def asan_lib():
    r = subprocess.run([os.path.join(STUB, "build.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "warning" not in r.stderr, r.stderr[-3000:]          # -Wall -Wextra clean
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("gcc has no libasan.so here")
    return os.path.join(STUB, "_build", "libvdyn_capi_asan.so"), libasan


@pytest.mark.parametrize("seed,threads", [(0, None), (1, "1"), (3, "8")])
def test_every_host_entry_point_under_asan_ubsan(asan_lib, seed, threads):
    lib, libasan = asan_lib
    env = dict(os.environ, LD_PRELOAD=libasan, HIPSTUB_SEED=str(seed),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("VDYN_COPY_THREADS", None)
    if threads:
        env["VDYN_COPY_THREADS"] = threads
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_host_layer_driver.py"), lib], capture_output=True,
                       text=True, env=env, cwd=REPO, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail and "hipstub:" not in tail, tail
    assert "checks passed" in r.stdout


def test_the_stub_exports_what_the_header_declares(asan_lib):
    """The sanitizer build is the SAME translation unit as the product's C ABI: every symbol include/vdyn.h declares."""
    import importlib
    import re
    lib, _ = asan_lib
    sys.path.insert(0, REPO)
    sigs = importlib.import_module("python-motionplanning_amd._lib").SIGNATURES
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r" T (vdyn_\w+)", out))
    assert set(sigs) <= have, sorted(set(sigs) - have)


def test_real_launchers_host_halves_under_asan_ubsan():
    """Level 2: csrc/vdyn_capi.hip AND both kernel translation units, compiled by hipcc --cuda-host-only (the real HIP
    headers, host code only) under the sanitizers, linked with the stub runtime (tests/hipstub/build_launchers.sh).  The
    launchers' host halves run for real -- make_dev_params, the long-double tire fits and their cache, the per-wheel /
    per-axle fit tables, fleet and candidate tables, LDS chunking, the choice of kernel instance -- for the default tires,
    one C per axle, four different C, a shape factor no fit covers, a negative stiffness, and a fleet mixing them.  A
    launch ends in the stub's hipLaunchKernel: block of 1..1024 threads, grid >= 1, dynamic LDS within the limit the
    launcher asked for (hipFuncSetAttribute) and the CU's 160 KB; every kernel family must have been launched."""
    hipcc = "/opt/rocm/bin/hipcc"
    rt = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"
    if not os.path.exists(hipcc) or not os.path.exists(rt):
        pytest.skip("needs ROCm's hipcc and clang's shared ASan runtime")
    r = subprocess.run([os.path.join(STUB, "build_launchers.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, HIPSTUB_LEVEL="2", HIPSTUB_SEED="1",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_host_layer_driver.py"),
                        os.path.join(STUB, "_build", "libvdyn_host_asan.so")], capture_output=True, text=True, env=env,
                       cwd=REPO, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail and "hipstub:" not in tail, tail
    assert "level 2:" in r.stdout and "distinct kernel instances" in r.stdout and "checks passed" in r.stdout


def test_staging_worker_threads_under_tsan():
    """The staging copies of the `_host` entry points run on worker threads (CopyPool, csrc/vdyn_capi.hip: a job posted
    under a mutex, slices claimed through an atomic counter, the caller copying too).  The same driver, the same stub,
    built with -fsanitize=thread and six copy threads: no data race may be reported."""
    r = subprocess.run([os.path.join(STUB, "build.sh"), "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    libtsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libtsan) or not os.path.exists(libtsan):
        pytest.skip("gcc has no libtsan.so here")
    env = dict(os.environ, LD_PRELOAD=libtsan, VDYN_COPY_THREADS="6", HIPSTUB_SEED="2", HIPSTUB_QUICK="1",
               TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0 exitcode=66")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_host_layer_driver.py"),
                        os.path.join(STUB, "_build", "libvdyn_capi_tsan.so")], capture_output=True, text=True, env=env,
                       cwd=REPO, timeout=1200)
    tail = (r.stdout + r.stderr)[-4000:]
    assert "ThreadSanitizer" not in r.stderr and "data race" not in r.stderr, tail
    assert r.returncode == 0 and "checks passed" in r.stdout, tail
