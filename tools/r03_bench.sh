#!/bin/bash
# usage: tools/r03_bench.sh <tag> [bench args...]  -> gpurun_out/r03/bench_<tag>.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
tag=$1; shift
timeout -k 10 900 python bench.py "$@" > gpurun_out/r03/bench_$tag.json 2> gpurun_out/r03/bench_$tag.err || { tail -20 gpurun_out/r03/bench_$tag.err; exit 1; }
python3 - <<PY
import json
d = json.loads(open("gpurun_out/r03/bench_$tag.json").read().strip().splitlines()[-1])
print("G edges/s %.2f  ms/step %.3f  frac %.4f  pipeline %s" % (d["value"] / 1e9, d["ms_per_step"], d["roofline"]["frac"], d["config"].get("pipeline")))
print(d["config"].get("placements_tried_ms_per_launch"))
print(d["config"].get("verified_batches"))
print({k: v for k, v in d.get("cpu_baseline", {}).items() if k != "sample"})
PY
