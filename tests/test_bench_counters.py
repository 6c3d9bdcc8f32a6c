"""CPU: bench.py ties the counter-derived fields of `roofline` to the code objects it loaded.

profiles/summarize.py stores vdyn_build_id() of the profiled library in the summary; bench.py compares it with the
library it loaded itself and reports `pmc_stale: true` -- without traffic or issue-slot figures -- on any mismatch."""
import importlib
import json
import os

import bench

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PMC = {"tag": "rXX", "build_id": "0123456789abcdef", "hbm_bytes_per_launch": 6.9e6,
       "valu_insts_per_wave_per_rk4_step": 372.4, "wave_cycles_per_valu_inst": 4.85}


def test_counters_of_the_loaded_build_are_reported():
    cf = bench.counter_fields(PMC, "0123456789abcdef", 65536 * 200, 0.153e-3)
    assert cf["pmc_stale"] is False and cf["traffic"] == 6.9e6
    assert cf["valu_insts_per_wave_step"] == 372.4 and cf["cycles_per_inst"] == 4.85
    assert 0.0 < cf["issue_frac"] < 1.0 and cf["pmc_source"].endswith("(rXX)")


def test_traffic_is_only_quoted_for_the_profiled_launch_shape():
    pmc = dict(PMC, grid=65536)
    assert bench.counter_fields(pmc, PMC["build_id"], 65536 * 200, 0.153e-3, rollouts=65536)["traffic"] == 6.9e6
    cf = bench.counter_fields(pmc, PMC["build_id"], 14000 * 50, 0.04e-3, rollouts=14000)
    assert cf["traffic"] is None and cf["pmc_stale"] is False
    assert cf["pmc_shape_mismatch"] == {"profiled_rollouts": 65536, "launched_rollouts": 14000}
    assert cf["valu_insts_per_wave_step"] == 372.4


def test_counters_of_another_build_are_dropped_and_flagged():
    for pmc, loaded in ((PMC, "fedcba9876543210"), ({k: v for k, v in PMC.items() if k != "build_id"}, "0123456789abcdef"),
                        (PMC, None)):
        cf = bench.counter_fields(pmc, loaded, 65536 * 200, 0.153e-3)
        assert cf["pmc_stale"] is True and cf["traffic"] is None
        assert "valu_insts_per_wave_step" not in cf and "cycles_per_inst" not in cf and "issue_frac" not in cf
    assert bench.counter_fields({}, "0123456789abcdef", 1, 1.0) == {"traffic": None}


def test_source_hash_follows_the_sources(tmp_path, monkeypatch):
    bld = importlib.import_module("python-motionplanning_amd._build")
    h0 = bld.source_hash()
    assert len(h0) == 16 and int(h0, 16) >= 0 and bld.source_hash() == h0
    assert bld.source_hash(extra_flags=("-DX",)) != h0
    # a copy of csrc with one byte more in one header hashes differently
    import shutil
    shutil.copytree(bld.CSRC, tmp_path / "csrc")
    with open(tmp_path / "csrc" / "vdyn_device.hpp", "a") as f:
        f.write("\n")
    monkeypatch.setattr(bld, "CSRC", str(tmp_path / "csrc"))
    assert bld.source_hash() != h0


def test_committed_summary_carries_a_build_id_or_is_treated_as_stale():
    with open(os.path.join(REPO, "profiles", "pmc_summary.json")) as f:
        pmc = json.load(f)
    cf = bench.counter_fields(pmc, pmc.get("build_id") or "x", 65536 * 200, 0.153e-3)
    assert cf["pmc_stale"] is (pmc.get("build_id") is None)
