#!/bin/bash
# HGT after a change: its parity tests, the cfg4 latency, kernels per call
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_hgt.py tests/test_gpu_random_sweep_hetero.py tests/test_gpu_fullsize_cfg45.py tests/test_gpu_examples.py tests/test_gpu_surface.py -x -q > gpurun_out/r03/tests_hgt.log 2>&1 || { tail -40 gpurun_out/r03/tests_hgt.log; exit 1; }
tail -3 gpurun_out/r03/tests_hgt.log
timeout -k 10 300 python tools/bench_hetero.py > gpurun_out/r03/bench_hetero.json 2> gpurun_out/r03/bench_hetero.err || { tail -20 gpurun_out/r03/bench_hetero.err; exit 1; }
cat gpurun_out/r03/bench_hetero.json
timeout -k 10 300 bash tools/trace_het.sh hgt 50
