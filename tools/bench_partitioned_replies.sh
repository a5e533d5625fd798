#!/bin/bash
# mode 2 with one rank: slot replies against compact replies, without an exchange and over RCCL with 1 / 2 / 3 lanes
# usage (on the GPU box): tools/bench_partitioned_replies.sh <round>    -> gpurun_out/<round>/partitioned_replies.jsonl
set -e -o pipefail
R=${1:-r04}
mkdir -p gpurun_out/$R
OUT=gpurun_out/$R/partitioned_replies.jsonl
: > $OUT
B="python bench.py --mode partitioned --batches-per-step 4096 --steps 12 --warmup 3"
for replies in auto compact; do
  $B --replies $replies | tail -1 >> $OUT
  for lanes in 1 2 3; do
    $B --replies $replies --force-exchange --lanes $lanes | tail -1 >> $OUT
  done
done
python - <<'PY' $OUT
import json, sys
for line in open(sys.argv[1]):
    r = json.loads(line)
    c = r["config"]
    print("%-8s lanes %d %-40s %6.2f G edges/s  %6.3f ms/call" % (c["replies"]["form"], c["lanes"],
          "exchange over RCCL" if "exchanging with itself" in c["parallelism"] else "no exchange", r["value"] / 1e9, r["ms_per_step"]))
PY
