#!/bin/bash
# rocprofv3 evidence for the kernels that are NOT the headline: one case of
# profiles/kernels_workload.py per run, kernel-trace + stats first, then every --pmc set in a run of
# its own (never mixed with a trace domain other than kernel-trace; the program goes directly
# after `--`).
# Usage (repo root, via gpurun):  bash profiles/collect_kernels.sh <tag> [case ...]
# Raw CSVs: gpurun_out/profk_<tag>/<case>/...; summary: gpurun_out/profk_<tag>/summary/<tag>_kernels_pmc_summary.json
set -eo pipefail
TAG=${1:-r02}
shift || true
CASES=${*:-"config2_f64_lane config2_f64_wheel config2_f64_axles config2_f64_four_c config5_mpc closed_loop closed_loop_datalog trajectory_dump per_rollout_controls spiral_lattice"}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/profk_$TAG
mkdir -p "$OUT/summary"
cd /tmp && export TMPDIR=/tmp
for CASE in $CASES; do
  D=$OUT/$CASE
  mkdir -p "$D"
  W="python3 $ROOT/profiles/kernels_workload.py --case $CASE"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$D/kt" -- $W --manifest "$D/manifest.json" > "$D/kt.log" 2>&1
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$D/pmc_fetch" -- $W > "$D/pmc_fetch.log" 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$D/pmc_write" -- $W > "$D/pmc_write.log" 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
      --output-format csv -d "$D/pmc_sq" -- $W > "$D/pmc_sq.log" 2>&1
  timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM \
      --output-format csv -d "$D/pmc_sq2" -- $W > "$D/pmc_sq2.log" 2>&1 || echo "pmc SQ2 failed for $CASE (non-fatal)"
  echo "$CASE done"
done
python3 "$ROOT/profiles/summarize_kernels.py" "$OUT" "$TAG"
