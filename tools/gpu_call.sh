#!/bin/bash
# One gpurun call: optional pytest files, optional same-process A/B of tuning variants, optional bench.py run.
# usage: tools/gpu_call.sh <round> <tag> [pytest args...] [-- ab_tuning variant specs...] [--- bench.py args...]
#   outputs under gpurun_out/<round>/: tests_<tag>.log, ab_<tag>.jsonl, bench_<tag>.json
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
rnd=$1; tag=$2; shift 2
out=gpurun_out/$rnd; mkdir -p "$out"
tests=(); ab=(); bench=(); mode=tests
for a in "$@"; do
  if [ "$a" == "--" ]; then mode=ab; continue; fi
  if [ "$a" == "---" ]; then mode=bench; continue; fi
  case $mode in tests) tests+=("$a");; ab) ab+=("$a");; bench) bench+=("$a");; esac
done
if [ ${#tests[@]} -gt 0 ]; then
  timeout -k 10 1000 python -m pytest "${tests[@]}" -x -q > "$out/tests_$tag.log" 2>&1 || { tail -60 "$out/tests_$tag.log"; exit 1; }
  tail -3 "$out/tests_$tag.log"
fi
if [ ${#ab[@]} -gt 0 ]; then
  args=(); for a in "${ab[@]}"; do [ "$a" != "default" ] && args+=("$a"); done
  SOL=${SOL:-0} timeout -k 10 700 python tools/ab_tuning.py "${args[@]}" > "$out/ab_$tag.jsonl" 2> "$out/ab_$tag.err" || { tail -30 "$out/ab_$tag.err"; exit 1; }
  python - "$out/ab_$tag.jsonl" <<'PY'
import json, sys
last = {}
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    if "round" in d:
        last[d["variant"]] = d
    elif "summary_ms_min" in d or "speed_of_light_of_the_output_contract_ms" in d:
        print(d)
for v, d in last.items():
    print(v, d["ms_per_launch"], d["roofline_frac"], d["stages_ms"])
PY
fi
if [ ${#bench[@]} -gt 0 ]; then
  args=(); for a in "${bench[@]}"; do [ "$a" != "default" ] && args+=("$a"); done
  timeout -k 10 1000 python bench.py "${args[@]}" > "$out/bench_$tag.json" 2> "$out/bench_$tag.err" || { tail -30 "$out/bench_$tag.err"; exit 1; }
  python - "$out/bench_$tag.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("G edges/s %.2f  ms/step %.3f  frac %.4f  pipeline %s" % (d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"], d["config"].get("pipeline")))
print(d["config"].get("placements_tried_ms_per_launch"))
print(d["config"].get("verified_batches"))
print({k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
PY
fi
