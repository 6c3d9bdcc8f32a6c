#!/usr/bin/env python3
"""Audit of gfx950 assembly (hipcc -S --cuda-device-only) for ONE code-generation hazard, found in round 5 as the
cause of round 4's shelved `RowReader` miscompare (profiles/README.md, "RowReader"):

    .LBB0_114:                          ; join block of `if (!ok) { SAFE step }` -- reached by the lanes that skipped
        v_accvgpr_write_b32 a109, v51   ;    the redo (s_cbranch_execz) AND by fall-through from the redo, exec = !ok
        v_accvgpr_write_b32 a107, v39   ; <- register-allocator copies of values live for ALL lanes (x, y) ...
        v_readlane_b32 s0, v255, 48     ; <- the saved exec mask, itself an SGPR spilled to a VGPR lane
        v_readlane_b32 s1, v255, 49
        s_or_b64 exec, exec, s[0:1]     ; <- ... placed BEFORE the exec mask is restored: lanes that did NOT take the
                                        ;    redo keep the previous step's x, y in a[106:109]

The allocator's copies / spills belong after the block's exec-restoring prologue.  When the restore's own operand has
to be reloaded from a spill lane first, this LLVM (ROCm 7.2) puts the copies at the very top of the block, in front of
the reload, i.e. under the PARTIAL exec mask of the fall-through predecessor.  Needs: SGPR pressure high enough that the
saved exec mask is spilled, and a divergent region whose join block receives allocator copies -- exactly the fp64 k = 12
rollout with twelve more scalar row offsets (RowReader) and lanes in the SAFE redo.

What is flagged: a JOIN block (two or more ways in, one of them an `s_cbranch_execz` -- the branch AROUND a divergent
region, whose target is by construction that region's join) whose FIRST instruction touching exec is `s_or_b64 exec, exec, <saved>` (an end-of-divergent-region restore) and that executes vector instructions (VALU
incl. v_accvgpr_*, VMEM, LDS) before it.  v_readlane / v_writelane (exec-independent; the spill reloads themselves) are
not counted.  A block that is only ever entered by `s_cbranch_execnz` or by falling through from an `s_and_saveexec`
and ends in a restore is the (possibly shared, tail-duplicated) BODY of a divergent `if`: it is meant to run under the
mask and is not flagged.

    python3 tools/isa/exec_restore_audit.py file.s [file2.s ...] [--json]     compiler assembly (hipcc -S)
    python3 tools/isa/exec_restore_audit.py --lib [libvdyn_hip.so] [--json]    the BUILT library: its gfx950 code
                                            objects, disassembled (llvm-objdump -d --symbolize-operands)

Basic blocks: a label (`.LBBn_m:`, `; %bb.m:`; in a disassembly `<Lm>:`) starts one, and so does the instruction after
any branch.  Exit code 1 when anything is flagged.  tests/test_isa_audit.py runs the --lib form on the shipped library
(so the hazard fails the build's tests rather than a result) and the file form on a reproducer of the flagged shape."""
import json
import os
import re
import subprocess
import sys
import tempfile

EXEC_WRITE = re.compile(r"^(s_\w+saveexec_b64\b|s_(mov|or|and|andn2|xor|orn2|not|cselect|wqm)_b64\s+exec\b|s_mov_b32\s+exec_(lo|hi)\b)")
END_CF = re.compile(r"^s_or_b64\s+exec,\s*exec,")
VECTOR = re.compile(r"^(v_|buffer_|global_|flat_|ds_|scratch_|tbuffer_|image_)")
LANE_OPS = ("v_readlane_b32", "v_writelane_b32")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True,
                             check=True).stdout.splitlines()
        return dict(zip(names, out))
    except (OSError, subprocess.CalledProcessError):
        return {n: n for n in names}


OBJ_LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:$")


BRANCH = re.compile(r"^(s_cbranch_\w+|s_branch)\s+(\S+)")
NO_FALLTHROUGH = ("s_branch", "s_setpc", "s_endpgm")


def audit_lines(lines):
    """[{kernel, block, line, preds, before: [instructions executed before the restore]}] for an iterable of assembly or
    disassembly lines.  Only JOIN blocks count (two or more ways in: branch targets + fall-through): a block with one
    way in that ends in a restore is the out-of-line body of a divergent `if` -- it is meant to run under the mask."""
    lines = [ln.rstrip("\n") for ln in lines]

    def parse(raw):
        """-> (kind, value): kernel / label / bb / end / ins / None"""
        s = raw.strip()
        if not s:
            return None, None
        m = OBJ_LABEL.match(s)
        if m:                                                   # llvm-objdump: `addr <name>:`
            return ("label", m.group(1)) if re.fullmatch(r"L\d+", m.group(1)) else ("kernel", m.group(1))
        if s.startswith("; %bb."):
            return "bb", s.split(":")[0].lstrip("; ")
        if s.startswith((";", "//")):
            return None, None
        m = re.match(r"^([A-Za-z_$][\w$.]*):", s)
        if m and not s.startswith(".L"):
            return "kernel", m.group(1)
        if s.startswith(".Lfunc_end"):
            return "end", None
        if s.startswith(".LBB"):
            return "label", s.split(":")[0]
        if s.startswith("."):
            return None, None
        ins = re.split(r";|//", s)[0].strip()
        return ("ins", ins) if ins else (None, None)

    parsed = [parse(ln) for ln in lines]
    # pass 1: how many branches target each label, per kernel (and how many of them are `s_cbranch_execz`: the branch
    # AROUND a divergent region when no lane enters it -- its target is that region's join block)
    targets, skips, kernel = {}, {}, None
    for kind, val in parsed:
        if kind == "kernel":
            kernel = val
        elif kind == "end":
            kernel = None
        elif kind == "ins" and kernel is not None:
            m = BRANCH.match(val)
            if m:
                key = (kernel, m.group(2))
                targets[key] = targets.get(key, 0) + 1
                if m.group(1) == "s_cbranch_execz":
                    skips[key] = skips.get(key, 0) + 1
    # pass 2
    findings, kernel, block, block_line, preds, skipped_to = [], None, None, 0, 0, 0
    pre, state, falls = [], "scan", False        # falls: control can fall through from the previous instruction
    kernels = set()
    for ln, (kind, val) in enumerate(parsed, 1):
        if kind is None:
            continue
        if kind == "kernel":
            kernel, block, block_line, pre, state, preds, falls = val, "entry", ln, [], "scan", 1, True
            continue
        if kind == "end":
            kernel = None
            continue
        if kernel is None:
            continue
        if kind == "label":
            block, block_line, pre, state = val, ln, [], "scan"
            preds = targets.get((kernel, val), 0) + (1 if falls else 0)
            skipped_to = skips.get((kernel, val), 0)
            falls = True
            continue
        if kind == "bb":
            block, block_line, pre, state, preds, skipped_to = val, ln, [], "scan", 1 if falls else 0, 0
            falls = True
            continue
        ins = val
        if ins.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            # what follows a branch is a new block with (so far) one way in: the fall-through of a conditional branch
            falls = not ins.startswith(NO_FALLTHROUGH)
            block, block_line, pre, state, preds, skipped_to = f"{block}+", ln, [], "scan", 1 if falls else 0, 0
            continue
        if state == "done":
            continue
        if EXEC_WRITE.match(ins):
            if END_CF.match(ins) and pre and preds >= 2 and skipped_to >= 1:
                findings.append({"kernel": kernel, "block": block, "line": ln, "block_line": block_line, "preds": preds,
                                 "restore": ins, "before": pre[:]})
                kernels.add(kernel)
            state = "done"
            continue
        if VECTOR.match(ins) and not ins.startswith(LANE_OPS):
            pre.append(ins)
    names = demangle(sorted(kernels))
    for fd in findings:
        fd["kernel_demangled"] = re.sub(r"\(.*$", "", names.get(fd["kernel"], fd["kernel"]))
    return findings


def audit(path):
    with open(path) as f:
        return audit_lines(f)


def audit_library(lib_path=None):
    """The same over every gfx950 code object inside the built library.  -> (findings, kernels seen)"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import code_object_meta as M
    if lib_path is None:
        sys.path.insert(0, M.ROOT)
        import importlib
        lib_path = importlib.import_module("python-motionplanning_amd._build").LIB_PATH
    findings, n_kernels = [], 0
    for i, img in enumerate(M.code_objects(lib_path)):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            proc = subprocess.Popen([os.path.join(M.LLVM, "llvm-objdump"), "-d", "--symbolize-operands",
                                     "--no-show-raw-insn", f.name], stdout=subprocess.PIPE, text=True)
            lines = []
            for ln in proc.stdout:
                if OBJ_LABEL.match(ln.strip()) and not re.match(r"^[0-9a-f]+ <L\d+>:$", ln.strip()):
                    n_kernels += 1
                lines.append(ln)
            proc.wait()
            if proc.returncode != 0:
                raise RuntimeError("llvm-objdump failed on a code object of " + lib_path)
        for fd in audit_lines(lines):
            fd["file"] = f"{os.path.basename(lib_path)}[code object {i}]"
            findings.append(fd)
    return findings, n_kernels


def main(argv):
    as_json = "--json" in argv
    files = [a for a in argv if not a.startswith("--")]
    allf = []
    if "--lib" in argv:
        allf, n = audit_library(files[0] if files else None)
        files = [f"{n} kernels"]
    elif not files:
        print(__doc__)
        return 2
    else:
        for p in files:
            for fd in audit(p):
                fd["file"] = p
                allf.append(fd)
    if as_json:
        print(json.dumps(allf, indent=1))
    else:
        for fd in allf:
            print(f"{fd['file']}:{fd['line']}: {fd['kernel_demangled']} block {fd['block']}: {len(fd['before'])} vector "
                  f"instruction(s) before `{fd['restore']}`: " + "; ".join(fd["before"][:6]) + (" ..." if len(fd["before"]) > 6 else ""))
        print(f"{len(allf)} block(s) flagged in " + (files[0] + " of the library" if "--lib" in argv else f"{len(files)} file(s)"))
    return 1 if allf else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
