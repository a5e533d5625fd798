#!/bin/bash
# rocprofv3 kernel stats of mode 2 with one rank: tools/prof_partitioned.sh <round> <replies: auto|compact> [tag] [more bench.py args]
set -e
R=$1; replies=${2:-auto}; tag=${3:-noexchange}; shift; shift; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$R/prof_part_${replies}_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $ROOT/bench.py --mode partitioned --replies $replies --batches-per-step 4096 --steps 6 --warmup 2 "$@" > $OUT/stats.log 2>&1
cd $ROOT
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/$R/partitioned_${replies}_${tag}_kernel_stats.csv \;
rm -rf $OUT/stats
tail -1 $OUT/stats.log | cut -c1-200
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$ROOT/gpurun_out/$R/partitioned_${replies}_${tag}_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %5s  avg %9.1f us  total %8.2f ms  %5s %%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
