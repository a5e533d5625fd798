#!/bin/bash
# PMC passes (L2 hit/miss, fabric read / write requests) of the windowed bench -> gpurun_out/pmc_<tag>.json
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-secondary --no-cpu-baseline --steps 3 --warmup 1 $@"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $ROOT/gpurun_out/pmc_${tag}_hit --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${tag}_hit.log 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum -d $ROOT/gpurun_out/pmc_${tag}_rd --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${tag}_rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum -d $ROOT/gpurun_out/pmc_${tag}_wr --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_${tag}_wr.log 2>&1
cd $ROOT
for k in hit rd wr; do python3 tools/pmc_minmax.py gpurun_out/pmc_${tag}_$k win_gather win_emit win_scatter win_hist ns_homo_uniform; done > gpurun_out/pmc_$tag.json
cat gpurun_out/pmc_$tag.json
