#!/bin/bash
# mode 2, one rank: the slot kernels' launch shapes by environment (A/B on one box)
# usage: tools/ab_partitioned_env.sh <round> "<bench args>" "VAR=val VAR=val" "VAR=val" ...   ("" = defaults)
R=$1; ARGS=$2; shift; shift
mkdir -p gpurun_out/$R
OUT=gpurun_out/$R/ab_partitioned_env.txt
for v in "$@"; do
  line=$(env $v python bench.py --mode partitioned --batches-per-step 4096 --steps 12 --warmup 3 $ARGS | tail -1)
  python - "$ARGS | $v" "$line" <<'PY' | tee -a $OUT
import json, sys
r = json.loads(sys.argv[2])
print("%-80s %6.2f G edges/s  %6.3f ms/call" % (sys.argv[1], r["value"] / 1e9, r["ms_per_step"]))
PY
done
