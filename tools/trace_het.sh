#!/bin/bash
# usage: tools/trace_het.sh <what> [calls] -> gpurun_out/trace_het_<what>.txt (kernels per call, busy time per call)
what=$1; calls=${2:-50}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_het_$what -- python3 $GRAFT_REPO_ROOT/tools/trace_het.py $what $calls > $GRAFT_REPO_ROOT/gpurun_out/trace_het_$what.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/trace_het_$what.txt
import csv, glob
calls = $calls + 1
f = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/trace_het_$what/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
skip = ("rmat", "csx", "coo", "radix", "ingest", "sort", "histogram")
tot_n = tot_ns = 0
print(open("$GRAFT_REPO_ROOT/gpurun_out/trace_het_$what.log").read().strip().splitlines()[-1])
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n, ns = int(r["Calls"]), float(r["TotalDurationNs"])
    if n < calls // 2:
        continue
    tot_n += n; tot_ns += ns
    print("%-70s calls/call %6.1f  us/call %8.1f  avg us %6.1f" % (r["Name"][:70], n / calls, ns / calls / 1e3, float(r["AverageNs"]) / 1e3))
print("TOTAL kernels per call %.1f, busy us per call %.1f" % (tot_n / calls, tot_ns / calls / 1e3))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/trace_het_$what.txt
