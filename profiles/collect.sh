#!/bin/bash
# Collect the rocprofv3 evidence for bench.py's dominant kernel on the GPU box:
#   1. --kernel-trace --stats            -> per-kernel average duration
#   2. --pmc passes (each on its own, never mixed with a trace domain other than
#      kernel-trace): FETCH_SIZE | WRITE_SIZE | SQ instruction/cycle counters
# Usage (from the repo root, via gpurun):  bash profiles/collect.sh <tag>
# Raw CSVs land in gpurun_out/prof_<tag>/, summaries in gpurun_out/prof_<tag>/summary/.
set -eo pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"

# the trace pass runs bench.py's default K = 200 / W = 20, so that the traced average is taken over the same
# warmed-up launches bench.py times; the counter passes need only a few launches
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extra > "$OUT/kt.log" 2>&1
echo "kernel-trace done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
echo "pmc FETCH_SIZE done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $BENCH > "$OUT/pmc_write.log" 2>&1
echo "pmc WRITE_SIZE done"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d "$OUT/pmc_sq" -- $BENCH > "$OUT/pmc_sq.log" 2>&1
echo "pmc SQ done"
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM \
    --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || echo "pmc SQ2 failed (non-fatal)"
python3 "$ROOT/profiles/summarize.py" "$OUT" "$TAG"
