#!/bin/bash
# usage: tools/sweep_bpl.sh "<batches-per-step> ..." [rounds] -> gpurun_out/sweep_bpl.log
out=$GRAFT_REPO_ROOT/gpurun_out/sweep_bpl.log
: > $out
for r in $(seq 1 ${2:-2}); do
for b in $1; do
  python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 6 --warmup 2 --batches-per-step $b 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$b', 'G edges/s %.2f' % (d['value']/1e9), 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['config']['batches_per_launch'])" >> $out || exit 1
done
done
cat $out
