#!/bin/bash
# round 3: the records kept under profiles/r03 (run through gpurun from the repo root; ~4 minutes)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r03
export TMPDIR=/tmp
for pl in push staged; do
  bash tools/profile_bench.sh r03_$pl windowed 16384 $pl > gpurun_out/r03/profile_$pl.log 2>&1 || { tail -5 gpurun_out/r03/profile_$pl.log; exit 1; }
done
SOL=1 timeout -k 10 400 python tools/ab_tuning.py "staged:staged=1" "push_r02_pipeline:fold_hist=0,fuse_first_hops=0" "push_fused_only:fold_hist=0" > gpurun_out/r03/ab_final.jsonl 2> gpurun_out/r03/ab_final.err || { tail gpurun_out/r03/ab_final.err; exit 1; }
timeout -k 10 200 tools/bin/probe_permute 26 > gpurun_out/r03/probe_permute.jsonl 2>/dev/null
timeout -k 10 600 python tools/bench_loader.py > gpurun_out/r03/loader.json 2> gpurun_out/r03/loader.err || { tail gpurun_out/r03/loader.err; exit 1; }
timeout -k 10 300 python tools/bench_hetero.py > gpurun_out/r03/hetero.json 2> gpurun_out/r03/hetero.err
timeout -k 10 300 python tools/bench_secondary.py > gpurun_out/r03/secondary_random_walk.json 2> gpurun_out/r03/secondary_random_walk.err
timeout -k 10 300 python tools/bench_misc.py > gpurun_out/r03/secondary_misc.json 2> gpurun_out/r03/secondary_misc.err
bash tools/trace_het.sh hgt 50 > /dev/null 2>&1; bash tools/trace_het.sh weighted 50 > /dev/null 2>&1; bash tools/trace_het.sh filtered 50 > /dev/null 2>&1; bash tools/trace_het.sh homo_weighted 30 > /dev/null 2>&1
python bench.py --mode partitioned --steps 6 --warmup 2 > gpurun_out/r03/part_w1.json 2> gpurun_out/r03/part_w1.err
python bench.py > gpurun_out/r03/bench_default.json 2> gpurun_out/r03/bench_default.err || { tail gpurun_out/r03/bench_default.err; exit 1; }
tail -c 600 gpurun_out/r03/ab_final.jsonl; python3 -c "
import json; d=json.loads(open('gpurun_out/r03/bench_default.json').read().strip().splitlines()[-1]); print(d['value']/1e9, d['ms_per_step'], d['roofline'], d['config']['pipeline'])"
