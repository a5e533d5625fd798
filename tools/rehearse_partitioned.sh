#!/bin/bash
# 2-rank rehearsal of bench.py --mode partitioned on a one-GPU box: both ranks use cuda:0, collectives over gloo
# (RCCL refuses two ranks on one device).  Records the JSON line; NOT a performance number for two GPUs.
export TG_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --mode partitioned --scale ${1:-22} --batches-per-step ${2:-256} --steps 3 --warmup 1
