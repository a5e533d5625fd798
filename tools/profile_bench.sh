#!/bin/bash
# Profiles the default bench.py run on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats           -> gpurun_out/prof/stats
#   2. two --pmc passes (read requests, write requests) of the same command -> gpurun_out/prof/pmc_{rd,wr}
#   3. tools/pmc_sum.py -> per-launch counter averages of the dominant kernel
# Copy what should be judged into profiles/ afterwards.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline $BENCH_ARGS"
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_rd --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc_wr --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_wr.log 2>&1
cd $ROOT
python3 tools/pmc_sum.py $OUT/pmc_rd ns_homo_uniform > $OUT/pmc_rd.json
python3 tools/pmc_sum.py $OUT/pmc_wr ns_homo_uniform > $OUT/pmc_wr.json
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
cat $OUT/pmc_rd.json $OUT/pmc_wr.json
head -5 $OUT/kernel_stats.csv
tail -1 $OUT/stats.log
