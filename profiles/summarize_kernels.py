#!/usr/bin/env python3
"""Condense profiles/collect_kernels.sh's CSVs into <out>/summary/<tag>_kernels_pmc_summary.json:
one entry per case of profiles/kernels_workload.py with the kernel's average duration (kernel
trace), registers / LDS, per-launch counter averages and the figures derived from them.

HBM bytes as profiles/summarize.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE
count KiB; on gfx950 FETCH_SIZE tallies 64 B per 128-B request of the streams it was calibrated
on, so reads are reported raw and x2 (upper bound) and `hbm_bytes_per_launch` uses the x2 figure."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, tag = sys.argv[1], sys.argv[2]
HBM_PEAK = 8.0e12


def find(base, sub, pattern):
    hits = glob.glob(os.path.join(base, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


def loaded_build_id():
    """vdyn_build_id() of the library these counters were taken on (the summary runs on the same box, right after
    the collection): what bench.py later compares with the library IT loaded."""
    import ctypes
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        lib = ctypes.CDLL(importlib.import_module("python-motionplanning_amd._build").LIB_PATH)
        lib.vdyn_build_id.restype = ctypes.c_char_p
        return lib.vdyn_build_id().decode()
    except (OSError, AttributeError):
        return None


def code_object_registers(rocprof_kernel_name):
    """vgpr / agpr / sgpr / scratch / spill counts of the kernel from the CODE OBJECT's metadata inside the built
    library (tools/isa/code_object_meta.py).  rocprofv3's VGPR_Count column is not that number on gfx950 (108 for the
    headline kernel whose code object says 212): it is kept under `rocprofv3_*` names only."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools", "isa"))
    try:
        import code_object_meta as M
        global _META
        try:
            _META
        except NameError:
            _META = M.kernel_meta()
        m = M.lookup(_META, rocprof_kernel_name)
    except Exception as e:                      # noqa: BLE001 -- the summary still carries the counters
        return {"code_object_registers_error": repr(e)}
    if m is None:
        return {"code_object_registers_error": "kernel not found in the library's code objects"}
    total = m["vgpr"] + m["agpr"]
    return {"vgpr": m["vgpr"], "agpr": m["agpr"], "sgpr": m["sgpr"], "scratch_bytes": m["scratch_bytes"],
            "vgpr_spills": m["vgpr_spills"], "sgpr_spills_to_vgpr_lanes": m["sgpr_spills"],
            # gfx950: 512 unified registers per SIMD lane, allocated in blocks of 8
            "occupancy_waves_per_simd": max(1, min(8, 512 // max(8, -(-total // 8) * 8))),
            "registers_source": "code object metadata (tools/isa/code_object_meta.py)"}


result = {"tag": tag, "build_id": loaded_build_id(), "cases": {}}
for mf in sorted(glob.glob(os.path.join(out, "*", "manifest.json"))):
    base = os.path.dirname(mf)
    man = json.load(open(mf))
    key = man["kernel"]
    s = dict(man)
    ks = find(base, "kt", "*kernel_stats.csv")
    if ks:
        for r in csv.DictReader(open(ks)):
            if key in r["Name"]:
                s.update(kernel_full=r["Name"].split("(")[0], calls=int(r["Calls"]), avg_ns=float(r["AverageNs"]),
                         min_ns=float(r["MinNs"]), max_ns=float(r["MaxNs"]))
    kt = find(base, "kt", "*kernel_trace.csv")
    if kt:
        durs = []
        for r in csv.DictReader(open(kt)):
            if key in r["Kernel_Name"]:
                durs.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                s.update(rocprofv3_vgpr_count=int(r["VGPR_Count"]), rocprofv3_accum_vgpr_count=int(r["Accum_VGPR_Count"]),
                         rocprofv3_sgpr_count=int(r["SGPR_Count"]),
                         static_lds_bytes=int(r["LDS_Block_Size"]), scratch_bytes=int(r["Scratch_Size"]),
                         grid=int(r["Grid_Size_X"]), workgroup=int(r["Workgroup_Size_X"]))
                if "registers_source" not in s:
                    s.update(code_object_registers(r["Kernel_Name"]))
        if durs:      # the timed launches are the last `launches` of the trace (after the warm-up loop)
            tail = durs[-int(man["launches"]):]
            s["timed_avg_ns"] = sum(tail) / len(tail)
    counters = defaultdict(list)
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
        cc = find(base, sub, "*counter_collection.csv")
        if not cc:
            continue
        per = defaultdict(float)
        for r in csv.DictReader(open(cc)):
            if key in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, name), v in per.items():
            counters[name].append(v)
    c = {k: sum(v) / len(v) for k, v in sorted(counters.items())}
    s["counters_per_launch"] = c
    dur = s.get("timed_avg_ns", s.get("avg_ns"))
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        s["fetch_bytes_raw"] = c["FETCH_SIZE"] * 1024
        s["fetch_bytes_x2_gfx950"] = c["FETCH_SIZE"] * 2048
        s["write_bytes"] = c["WRITE_SIZE"] * 1024
        s["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 2048 + c["WRITE_SIZE"] * 1024
        s["hbm_bytes_over_algorithmic"] = s["hbm_bytes_per_launch"] / man["algo_bytes"]
        if dur:
            s["hbm_GBs_counter"] = s["hbm_bytes_per_launch"] / dur          # bytes / ns = GB/s
            s["hbm_frac_counter"] = s["hbm_bytes_per_launch"] / (dur * 1e-9) / HBM_PEAK
            s["hbm_GBs_algorithmic"] = man["algo_bytes"] / dur
            s["hbm_frac_algorithmic"] = man["algo_bytes"] / (dur * 1e-9) / HBM_PEAK
    if c.get("SQ_WAVES") and "SQ_INSTS_VALU" in c:
        s["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        # per RK4 step of one rollout chain: lanes that integrate several rollouts in turn (MPC) do more steps per wave
        steps_per_wave = man["vehicle_steps"] / (c["SQ_WAVES"] * 64.0) if "quad" not in key else man["steps_per_lane"]
        s["rk4_steps_per_wave"] = steps_per_wave
        s["valu_insts_per_wave_per_rk4_step"] = s["valu_insts_per_wave"] / steps_per_wave
        if "SQ_WAVE_CYCLES" in c:
            s["wave_cycles_per_valu_inst"] = 4.0 * c["SQ_WAVE_CYCLES"] / c["SQ_INSTS_VALU"]
        if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
            s["wait_any_frac_of_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if dur:
        s["steps_per_s"] = man["vehicle_steps"] / (dur * 1e-9)
    if "GRBM_GUI_ACTIVE" in c and dur:
        s["clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / dur       # 8 XCDs tick the counter
    result["cases"][man["case"]] = s
with open(os.path.join(out, "summary", f"{tag}_kernels_pmc_summary.json"), "w") as f:
    json.dump(result, f, indent=1)
for k, s in result["cases"].items():
    print(k, {x: s.get(x) for x in ("timed_avg_ns", "vgpr", "agpr", "valu_insts_per_wave_per_rk4_step", "wave_cycles_per_valu_inst",
                                    "wait_any_frac_of_wave_cycles", "hbm_bytes_per_launch", "hbm_frac_counter",
                                    "hbm_bytes_over_algorithmic", "steps_per_s")})
