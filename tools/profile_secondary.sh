#!/bin/bash
# rocprofv3 --kernel-trace --stats of the secondary benches (run through gpurun from the repo root); the per-kernel
# summaries land in gpurun_out/prof2/<name>_kernel_stats.csv -- copy what should be judged into profiles/.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for name in bench_hetero bench_scan bench_secondary bench_gather bench_misc bench_partitioned; do
  BATCHES_ENV=""
  rocprofv3 --kernel-trace --stats -d $OUT/$name --output-format csv -- python3 $ROOT/tools/$name.py > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; exit 1; }
  find $OUT/$name -name "*kernel_stats.csv" -exec cp {} $OUT/${name}_kernel_stats.csv \;
  rm -rf $OUT/$name    # the traces are large; only the summaries travel back
  echo "$name done"
done
