"""Build recipe of ``libvdyn_hip.so`` (explicit hipcc, in-tree, gfx950 only).

``python -m python_motionplanning_amd._build`` or ``__graft_entry__.build()``.
hipcc cross-compiles without a GPU; the built library is git-ignored but
travels to the GPU box with the working tree.
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# VDYN_LIB_PATH: load/build an alternative in-tree build (A/B experiments only)
LIB_PATH = os.environ.get("VDYN_LIB_PATH") or os.path.join(PKG_DIR, "libvdyn_hip.so")
# One wave per SIMD is the operating point of these kernels: schedule for instruction-level
# parallelism, not for occupancy.  A/B at the sustained clock (three runs each, +-0.2 %): the fp32
# kernels are fastest under max-ilp (headline -1.7 % against the default strategy), the fp64 ones
# under iterative-ilp (configs[1] 1.02 -> 0.86 ms) -- hence one translation unit per precision.  Re-measured with
# the fitted tire chain (round 2, one box): headline max-ilp 0.1536-0.1542 ms, iterative-ilp 0.1542, default
# (max-occupancy) 0.1548, iterative-minreg 0.1713; closed loop 0.317 / 0.330 / 0.313 / 0.390.
SOURCES = [("vdyn_kernels_f64_rollout.hip", ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]),     # the longest first
           ("vdyn_kernels_f32_rollout.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
           ("vdyn_kernels_f64_rest.hip", ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]),
           ("vdyn_kernels_f32_rest.hip", ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
           ("vdyn_capi.hip", [])]
HEADERS = [os.path.join(CSRC, h) for h in ("vdyn_kernels.hip", "vdyn_device.hpp", "vdyn_internal.hpp",
                                           "vdyn_fastmath.hpp", "vdyn_packed.hpp", "vdyn_controls.hpp",
                                           "vdyn_quad.hpp", "vdyn_quad_packed.hpp", "vdyn_lattice.hpp")] + \
          [os.path.join(PKG_DIR, os.pardir, "include", "vdyn.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
               "-fno-fast-math", "-Wall", "-Wno-unused-function",
               # the packed fp32 step is written by hand (csrc/vdyn_packed.hpp); the SLP vectoriser
               # pairs unrelated scalars and pays more v_mov shuffles than it saves: measured
               # 5 % slower on the scalar step, 8.5 % slower on the packed one (0.2556 vs 0.2356 ms)
               "-fno-slp-vectorize"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def source_hash(extra_flags=()):
    """First 16 hex digits of the SHA-256 of everything the code objects depend on: every file under csrc/, the
    public header and the compiler flags.  Compiled into the library (vdyn_build_id) so that a counter summary under
    profiles/ can be tied to the build it was taken on."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    files.append(os.path.normpath(os.path.join(PKG_DIR, os.pardir, "include", "vdyn.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(repr((HIPCC_FLAGS, SOURCES, tuple(extra_flags))).encode())
    return h.hexdigest()[:16]


def _read_id():
    """(build id, extra flags) the library beside LIB_PATH was built with; (None, ()) when there is no record."""
    try:
        lines = open(LIB_PATH + ".id").read().splitlines()
    except OSError:
        return None, ()
    extra = tuple(json.loads(lines[1])) if len(lines) > 1 and lines[1].strip() else ()
    return (lines[0].strip() if lines else None), extra


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    # the id the library was built with (written beside it at link time, together with any extra flags of that build)
    # against the sources as they are now: catches an edit made WHILE a build was running, which modification times
    # alone miss.  The hash covers every source, header and flag: a file that is merely NEWER than the library (a
    # checkout, a copy onto the GPU box) with the same content is not a reason to compile for four minutes.
    built, extra = _read_id()
    # a library built once with extra flags (a diagnostic -D, phase timers) is never "fresh" for a plain build() / load():
    # tests and bench would run a non-default build silently (its id says so -- source_hash covers the flags -- but
    # nothing would rebuild it)
    return built is None or bool(extra) or built != source_hash(())


def _content_hash(paths):
    import hashlib
    h = hashlib.sha256()
    for f in paths:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def build(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source of the package (in parallel, one object each) and link
    libvdyn_hip.so."""
    extra_flags = tuple(extra_flags)
    if not force and not extra_flags and not is_stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = hipcc_path()
    objdir = os.path.join(PKG_DIR, "build", os.path.basename(LIB_PATH) + ".d")
    os.makedirs(objdir, exist_ok=True)
    build_id = source_hash(extra_flags)

    def deps_of(src):
        # the C ABI unit sees the launchers' declarations only; the kernel units see every header
        if src == "vdyn_capi.hip":
            return [os.path.join(CSRC, src), os.path.join(CSRC, "vdyn_internal.hpp"), HEADERS[-1]]
        return [os.path.join(CSRC, src)] + HEADERS

    def compile_one(item):
        src, flags = item
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, *HIPCC_FLAGS, *flags, *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if src == "vdyn_capi.hip":
            cmd.insert(1, f'-DVDYN_BUILD_ID="{build_id}"')
        # an object is reused when its command line AND the content of everything it was compiled from are what its
        # stamp records.  The content hash is taken BEFORE compiling: a source edited while its unit compiles leaves
        # a stamp that no longer matches, whatever the modification times say.
        stamp = obj + ".cmd"
        want = " ".join(cmd) + "\n" + _content_hash(deps_of(src))
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
            return obj
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.path.exists(stamp):
            os.remove(stamp)
        res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout)
        with open(stamp + ".tmp", "w") as f:
            f.write(want)
        os.replace(stamp + ".tmp", stamp)
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as pool:
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH + ".tmp", *objs]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout)
    # the id first goes stale-side: a crash between the two renames leaves a library whose id does not match (rebuilt
    # next time), never a new library under the old id
    if os.path.exists(LIB_PATH + ".id"):
        os.remove(LIB_PATH + ".id")
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    with open(LIB_PATH + ".id.tmp", "w") as f:
        f.write(build_id + "\n" + (json.dumps(list(extra_flags)) if extra_flags else "") + "\n")
    os.replace(LIB_PATH + ".id.tmp", LIB_PATH + ".id")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
