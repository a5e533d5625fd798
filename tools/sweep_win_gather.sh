#!/bin/bash
# usage: tools/sweep_win_gather.sh "<threads>x<blocks> ..."  -> gpurun_out/sweep_gather.log
# the window-ordered launch of bench.py under different geometries of the gather kernel (TG_WIN_GATHER_THREADS/_BLOCKS)
out=$GRAFT_REPO_ROOT/gpurun_out/sweep_gather.log
: > $out
for cfg in $1; do
  t=${cfg%x*}; b=${cfg#*x}
  TG_WIN_GATHER_THREADS=$t TG_WIN_GATHER_BLOCKS=$b python3 $GRAFT_REPO_ROOT/bench.py --form windowed --no-secondary --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg', 'G edges/s %.2f' % (d['value']/1e9), 'ms/step %.3f' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'])" >> $out || exit 1
done
cat $out
