#!/bin/bash
# usage: tools/r03_call.sh <tag> [pytest files...] -- [ab_tuning variant specs...]
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
tag=$1; shift
tests=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do tests+=("$1"); shift; done; [ "$1" == "--" ] && shift
if [ ${#tests[@]} -gt 0 ]; then
  timeout -k 10 900 python -m pytest "${tests[@]}" -x -q > gpurun_out/r03/tests_$tag.log 2>&1 || { tail -40 gpurun_out/r03/tests_$tag.log; exit 1; }
  tail -3 gpurun_out/r03/tests_$tag.log
fi
if [ $# -gt 0 ]; then
  SOL=${SOL:-0} timeout -k 10 600 python tools/ab_tuning.py "$@" > gpurun_out/r03/ab_$tag.jsonl 2> gpurun_out/r03/ab_$tag.err || { tail -20 gpurun_out/r03/ab_$tag.err; exit 1; }
  python - <<PY
import json
for ln in open("gpurun_out/r03/ab_$tag.jsonl"):
    d = json.loads(ln)
    if "round" in d and d["round"] == 2:
        print(d["variant"], d["ms_per_launch"], d["roofline_frac"], d["stages_ms"])
    elif "summary_ms_min" in d or "speed_of_light_of_the_output_contract_ms" in d:
        print(d)
PY
fi
