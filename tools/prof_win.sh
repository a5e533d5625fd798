#!/bin/bash
# usage: tools/prof_win.sh <tag> [bench args...]   (env passes through) -> gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 4 --warmup 2 "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$tag/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "tg::win" in r["Name"] or "ns_homo_uniform" in r["Name"]:
        print("$tag", r["Name"][:40], r["Calls"], "avg %.1f us min %.1f max %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
