#!/bin/bash
# One-rank rehearsal of bench.py's exchange step on the GPU box: the default-K bench line
# without a collective, with the RCCL all-gather overlapped / not overlapped, and with the peer-copy exchange -- all on
# the `nccl` backend, one rank, one GPU (--force-collective).  Each run is a fresh process.
# Usage (repo root, via gpurun):  bash tools/exchange_one_rank.sh <tag>   ->  gpurun_out/<tag>_exchange_one_rank.json
set -eo pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
COMMON="--no-extra --no-cpu-baseline --no-sections"
run() {   # name, extra flags
  local name=$1; shift
  timeout -k 10 300 python3 "$ROOT/bench.py" $COMMON "$@" > "$OUT/xchg_$name.json" 2> "$OUT/xchg_$name.err"
  echo "$name done"
}
run none
run rccl_overlapped --force-collective --exchange rccl
run rccl_no_overlap --force-collective --exchange rccl --no-overlap
run p2p_overlapped --force-collective --exchange p2p
run p2p_no_overlap --force-collective --exchange p2p --no-overlap
run none_again
python3 - "$OUT" "$TAG" <<'PY'
import json, sys, os
out, tag = sys.argv[1], sys.argv[2]
keep = ("value", "ms_per_step", "steps", "warmup", "n_gpus", "world_seen", "dist_backend", "exchange", "build_id",
        "host_enqueue_ms_per_step", "value_excluding_collective")
res = {"tag": tag, "what": "bench.py default K / W, one rank on one MI355X, nccl backend (--force-collective)", "runs": {}}
for name in ("none", "rccl_overlapped", "rccl_no_overlap", "p2p_overlapped", "p2p_no_overlap", "none_again"):
    line = [l for l in open(os.path.join(out, f"xchg_{name}.json")) if l.startswith("{")][-1]
    j = json.loads(line)
    res["runs"][name] = {k: j.get(k) for k in keep}
    res["runs"][name]["kernel_ms"] = j["roofline"]["kernel_ms"]
base = res["runs"]["none"]["ms_per_step"]
for name, r in res["runs"].items():
    r["ms_over_no_collective"] = r["ms_per_step"] / base
json.dump(res, open(os.path.join(out, f"{tag}_exchange_one_rank.json"), "w"), indent=1)
print(json.dumps({k: (round(v["ms_per_step"], 4), round(v["ms_over_no_collective"], 3)) for k, v in res["runs"].items()}))
PY
