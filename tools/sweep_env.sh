#!/bin/bash
# usage: tools/sweep_env.sh "<ENV=a> <ENV=b> ..." [rounds] [bench args...] -> gpurun_out/sweep_env.log
# the default bench launch under several settings of one tuning knob, interleaved, on one box
list=$1; rounds=${2:-2}; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/sweep_env.log
: > $out
for r in $(seq 1 $rounds); do
  for e in $list; do
    env $e python3 $GRAFT_REPO_ROOT/bench.py --no-secondary --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>/dev/null \
      | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$e', 'G edges/s %.2f' % (d['value']/1e9), 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])" >> $out || exit 1
  done
done
cat $out
