#!/bin/bash
# Profiles the default bench.py launch on the GPU box (run through gpurun from the repo root):
#   tools/profile_bench.sh <tag> <form: windowed|fused> <batches-per-step> [pipeline: push|staged]
#   1. rocprofv3 --kernel-trace --stats                       -> gpurun_out/prof_<tag>/stats (+ kernel_stats.csv)
#   2. two --pmc passes (read requests, write requests) of the same command
#   3. tools/pmc_traffic.py -> gpurun_out/prof_<tag>/pmc_traffic.json (bytes per launch, all kernels of the launch)
# Copy what should be judged into profiles/ afterwards.
set -e
tag=$1; form=$2; bpl=$3; pl=${4:-push}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-secondary --no-verify --placements 1 --form $form --pipelines $pl --batches-per-step $bpl --steps 6 --warmup 2"
rocprofv3 --kernel-trace --stats -d $OUT/stats --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $OUT/pmc_rd --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $OUT/pmc_wr --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_wr.log 2>&1
cd $ROOT
python3 tools/pmc_traffic.py $OUT/pmc_rd $OUT/pmc_wr $form $bpl $pl > $OUT/pmc_traffic.json
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/pmc_rd $OUT/pmc_wr
cat $OUT/pmc_traffic.json
head -12 $OUT/kernel_stats.csv | cut -c1-160
tail -1 $OUT/stats.log | cut -c1-400
