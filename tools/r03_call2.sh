#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_windowed_timed_scale.py tests/test_gpu_ns_homo_windowed.py tests/test_gpu_random_sweep_windowed.py tests/test_gpu_partitioned.py -x -q > gpurun_out/r03/tests_windowed.log 2>&1 || { tail -40 gpurun_out/r03/tests_windowed.log; exit 1; }
tail -3 gpurun_out/r03/tests_windowed.log
SOL=0 timeout -k 10 400 python tools/ab_tuning.py "push:staged=0" > gpurun_out/r03/ab_tuning_2.jsonl 2> gpurun_out/r03/ab_tuning_2.err || { tail -20 gpurun_out/r03/ab_tuning_2.err; exit 1; }
cat gpurun_out/r03/ab_tuning_2.jsonl
