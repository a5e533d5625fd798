#!/bin/bash
# round 3, first GPU call: new parity tests, the permutation probe, tuning A/B + speed of light
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03
timeout -k 10 500 python -m pytest tests/test_gpu_windowed_timed_scale.py tests/test_gpu_ns_homo_windowed.py tests/test_gpu_random_sweep_windowed.py -x -q > gpurun_out/r03/tests_windowed.log 2>&1 || { tail -30 gpurun_out/r03/tests_windowed.log; exit 1; }
tail -3 gpurun_out/r03/tests_windowed.log
timeout -k 10 200 tools/bin/probe_permute 26 > gpurun_out/r03/probe_permute.jsonl 2> gpurun_out/r03/probe_permute.err || { cat gpurun_out/r03/probe_permute.err; exit 1; }
cat gpurun_out/r03/probe_permute.jsonl
timeout -k 10 400 python tools/ab_tuning.py "r02_pipeline:fold_hist=0,fuse_first_hops=0" "fused_first_hops_only:fold_hist=0" "fold_512:emit_blocks=512" "fold_1024:emit_blocks=1024" > gpurun_out/r03/ab_tuning_1.jsonl 2> gpurun_out/r03/ab_tuning_1.err || { tail -20 gpurun_out/r03/ab_tuning_1.err; exit 1; }
cat gpurun_out/r03/ab_tuning_1.jsonl
