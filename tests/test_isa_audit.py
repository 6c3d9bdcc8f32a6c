"""CPU: every kernel of the BUILT library, disassembled and held to one rule -- no vector instruction in front of an exec
restore at the join block of a divergent region (tools/isa/exec_restore_audit.py).

That shape is the cause of round 4's shelved "RowReader" miscompare, found in round 5 from the ISA: this compiler (ROCm
7.2's clang-22) can place register-allocator copies of loop-carried values at the very top of a join block, in front of
the `s_or_b64 exec, exec, <saved>` that ends the region, when scalar copies / spill reloads sit between the block's
start and that restore.  The copies then run under the region's PARTIAL mask: lanes that skipped the region keep a
stale value.  It hit the fp64 k = 12 rollout -- x, y parked in AGPRs, lanes beside a SAFE-redo lane kept the previous
step's position -- once with RowReader (never shipped) and once in the SHIPPED shared-table general-chain instance
(`rollout_kernel<double, 12, 1, false, false, false>`, a handle whose tire fit is refused; no test reached it with SAFE
lanes until tests/test_gpu_parity.py::test_general_chain_with_safe_redo_lanes).  Since round 5 the lane kernels' redo is
not a divergent region at all (rk4_advance, csrc/vdyn_device.hpp); this test keeps every OTHER divergent region of
every instance honest, so that the hazard fails the build's tests rather than a result."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools", "isa"))

BUGGY = """\
_Z6kernelv:
	s_and_saveexec_b64 s[4:5], vcc
	s_cbranch_execz .LBB0_2
; %bb.1:
	v_mov_b32_e32 v38, v112
.LBB0_2:
	v_accvgpr_write_b32 a106, v38
	s_mov_b32 s78, s80
	v_readlane_b32 s0, v255, 48
	v_readlane_b32 s1, v255, 49
	s_or_b64 exec, exec, s[0:1]
	s_endpgm
.Lfunc_end0:
"""
# the out-of-line BODY of a divergent `if`, entered by s_cbranch_execnz and (tail-duplicated) by fall-through from another
# s_and_saveexec: meant to run under the mask -- and a proper join, restore first
FINE = """\
_Z6kernelv:
	s_and_saveexec_b64 s[8:9], vcc
	s_cbranch_execnz .LBB0_3
.LBB0_1:
	s_or_b64 exec, exec, s[8:9]
	v_accvgpr_write_b32 a106, v38
	s_and_saveexec_b64 s[8:9], vcc
	s_cbranch_execz .LBB0_4
.LBB0_3:
	v_mul_f32_e32 v6, v1, v6
	ds_write_b64 v8, v[6:7]
	s_or_b64 exec, exec, s[8:9]
	s_branch .LBB0_1
.LBB0_4:
	s_or_b64 exec, exec, s[8:9]
	v_accvgpr_write_b32 a107, v39
	s_endpgm
.Lfunc_end0:
"""


def test_audit_flags_the_hazard_and_not_its_lookalikes(tmp_path):
    import exec_restore_audit as A
    f = A.audit_lines(BUGGY.splitlines())
    assert len(f) == 1 and f[0]["block"] == ".LBB0_2" and f[0]["before"] == ["v_accvgpr_write_b32 a106, v38"]
    assert f[0]["restore"].startswith("s_or_b64 exec, exec, s[0:1]")
    assert A.audit_lines(FINE.splitlines()) == []
    # the same through llvm-objdump's form (labels `<L0>:`, operands symbolised)
    dis = BUGGY.replace("_Z6kernelv:", "0000000000001900 <_Z6kernelv>:").replace(".LBB0_2:", "0000000000001a00 <L0>:") \
               .replace("s_cbranch_execz .LBB0_2", "s_cbranch_execz L0").replace("; %bb.1:\n", "").replace(".Lfunc_end0:\n", "")
    f = A.audit_lines(dis.splitlines())
    assert len(f) == 1 and f[0]["block"] == "L0"


def test_no_shipped_kernel_has_a_vector_instruction_in_front_of_an_exec_restore_at_a_join():
    import importlib
    import exec_restore_audit as A
    sys.path.insert(0, REPO)
    bld = importlib.import_module("python-motionplanning_amd._build")
    bld.build()
    findings, n_kernels = A.audit_library(bld.LIB_PATH)
    assert n_kernels >= 200, f"only {n_kernels} kernels found in {bld.LIB_PATH}"
    assert findings == [], "\n".join(f"{fd['kernel_demangled']} block {fd['block']}: {fd['before'][:4]} before `{fd['restore']}`"
                                     for fd in findings)


def test_reproducer_is_flagged_only_with_both_diagnostic_macros(tmp_path):
    """tools/isa/reader_all.hip: RowReader for every k + the divergent form of the redo, in the fp64 k = 12 instance =
    round 4's failing variant.  It must be flagged (the tool sees the real thing); the shipped configuration of the same
    instance must not."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import exec_restore_audit as A
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm",
            "-amdgpu-sched-strategy=iterative-ilp", "-I", os.path.join(REPO, "python-motionplanning_amd", "csrc"), "-S",
            "--cuda-device-only"]
    src = os.path.join(REPO, "tools", "isa", "reader_all.hip")
    for flags, want in ((["-DVDYN_READER_ALL", "-DVDYN_MASKED_REDO"], 1), ([], 0)):
        out = str(tmp_path / f"ra{want}.s")
        r = subprocess.run(base + flags + ["-o", out, src], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        f = A.audit(out)
        assert len(f) == want, (flags, f)
        if want:
            assert all(i.startswith("v_accvgpr_write_b32") for i in f[0]["before"]) and "rollout_kernel" in f[0]["kernel"]
