#!/bin/bash
# Kernel timeline of the window-ordered launch under one tuning variant (rocprofv3 --kernel-trace around tools/ab_tuning.py
# with one round): which kernels overlap on the two streams, and what each takes beside the other.
#   tools/prof_timeline.sh <round> <tag> "<variant spec: name:key=val,...>" [launch index from the end, default 3]
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
rnd=$1; tag=$2; spec=$3; back=${4:-3}
OUT=$ROOT/gpurun_out/$rnd/tl_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ROUNDS=1 SOL=0 G=${G:-16384} rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/tools/ab_tuning.py "$spec" > $OUT/ab.jsonl 2> $OUT/ab.err || { tail -20 $OUT/ab.err; exit 1; }
cd $ROOT
python3 - $OUT $back <<'PY' > $OUT/timeline.txt
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "win_vtab_kernel" in r[2] or "win_first_hops" in r[2]]
which = len(starts) - int(sys.argv[2])
lo, hi = starts[which], starts[which + 1]
t0 = rows[lo][0]
for s, e, name, q in rows[lo:hi]:
    short = name.split("(")[0].replace("void tg::", "")[:48]
    print("%9.1f %9.1f  %8.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
print("launch span %.1f us" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3))
PY
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
cat $OUT/timeline.txt
