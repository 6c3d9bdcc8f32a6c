#!/usr/bin/env python3
"""Register allocation of every kernel AS THE CODE OBJECT RECORDS IT, read from the built library itself (seconds; no
recompilation): the gfx950 code objects are cut out of libvdyn_hip.so's `.hip_fatbin` section (clang offload bundles,
one per translation unit) and their NT_AMDGPU_METADATA notes are printed by llvm-readelf.

    python3 tools/isa/code_object_meta.py [kernel-substring] [--json]

rocprofv3's `VGPR_Count` column is NOT this number on gfx950 (it reports 108 for the headline kernel whose code object
says `.vgpr_count: 212`): profiles/summarize*.py take vgpr / agpr / sgpr / spills from here and keep rocprofv3's column
under its own name."""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """The gfx950 ELF images inside `lib_path`."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib_path,
                        os.path.join(td, "discard.so")], check=True, capture_output=True)
        data = open(fat, "rb").read()
    out, pos = [], 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return out
        n = struct.unpack_from("<Q", data, i + len(MAGIC))[0]
        off = i + len(MAGIC) + 8
        for _ in range(n):
            o, size, ln = struct.unpack_from("<QQQ", data, off)
            off += 24
            target = data[off:off + ln].decode()
            off += ln
            if "gfx950" in target and size:
                out.append(data[i + o:i + o + size])
        pos = i + 1


def kernel_meta(lib_path=None):
    """{demangled kernel name (template arguments kept, parameter list cut): {vgpr, agpr, sgpr, scratch_bytes,
    vgpr_spills, sgpr_spills, static_lds_bytes, max_flat_workgroup_size}} for every kernel of the library."""
    if lib_path is None:
        sys.path.insert(0, ROOT)
        import importlib
        lib_path = importlib.import_module("python-motionplanning_amd._build").LIB_PATH
    res = {}
    for img in code_objects(lib_path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], check=True,
                                 capture_output=True, text=True).stdout
        # the YAML document sits between "---" and "..." (its body at column 0)
        m = re.search(r"AMDGPU Metadata:\n\s*---\n(.*?)\n\.\.\.", txt, re.S)
        md = (yaml.load(m.group(1), Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader)) if m else None) or {}
        ks = md.get("amdhsa.kernels", [])
        names = subprocess.run(["c++filt"], input="\n".join(k[".name"] for k in ks), capture_output=True,
                               text=True).stdout.split("\n")
        for k, n in zip(ks, names):
            # "void vdyn::kernel<...>(args)" -> "vdyn::kernel<...>"
            n = re.sub(r"^void ", "", n)
            depth, cut = 0, len(n)
            for j, ch in enumerate(n):
                depth += ch == "<"
                depth -= ch == ">"
                if ch == "(" and depth == 0:
                    cut = j
                    break
            res[n[:cut]] = {"vgpr": k[".vgpr_count"], "agpr": k.get(".agpr_count", 0), "sgpr": k[".sgpr_count"],
                            "scratch_bytes": k[".private_segment_fixed_size"], "vgpr_spills": k[".vgpr_spill_count"],
                            "sgpr_spills": k[".sgpr_spill_count"], "static_lds_bytes": k[".group_segment_fixed_size"],
                            "max_flat_workgroup_size": k[".max_flat_workgroup_size"]}
    return res


def lookup(meta, rocprof_name):
    """The entry for a kernel as rocprofv3 names it ("void vdyn::k<...>(...)" or truncated): exact match on the
    name up to the parameter list, else the unique entry the name starts with / that starts with the name."""
    n = re.sub(r"^void ", "", rocprof_name).split("(")[0].strip()
    if n in meta:
        return meta[n]
    hits = [v for k, v in meta.items() if k.startswith(n) or n.startswith(k)]
    return hits[0] if len(hits) == 1 else None


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    want = args[0] if args else ""
    meta = {k: v for k, v in kernel_meta().items() if want in k}
    if "--json" in sys.argv:
        print(json.dumps(meta, indent=1))
    else:
        print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>7} {'vspill':>6} {'sspill':>6} {'lds':>6}  kernel")
        for k, v in sorted(meta.items()):
            print(f"{v['vgpr']:5d} {v['agpr']:5d} {v['sgpr']:5d} {v['scratch_bytes']:7d} {v['vgpr_spills']:6d} "
                  f"{v['sgpr_spills']:6d} {v['static_lds_bytes']:6d}  {k[:150]}")
