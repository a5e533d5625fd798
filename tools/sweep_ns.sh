#!/bin/bash
for cfg in "0 0" "1 0" "1 1" "0 1" "1 1" "0 0"; do
  set -- $cfg
  r=$(python bench.py --steps 8192 --warmup 1024 --no-cpu-baseline --idx32 $1 --ptr32 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['roofline']['avg_launch_ms'],4), round(d['value']/1e9,2))")
  echo "idx32=$1 ptr32=$2  ->  $r"
done
