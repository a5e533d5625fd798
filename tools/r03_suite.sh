#!/bin/bash
# the whole -m gpu suite in one process + the CPU-side check of the same tree
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03/suite_gpu.log 2>&1 || { tail -40 gpurun_out/r03/suite_gpu.log; exit 1; }
tail -3 gpurun_out/r03/suite_gpu.log
