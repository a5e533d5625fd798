#!/bin/bash
# A/B sweep of the headline kernel's tuning knobs on ONE box (boxes differ by 10-20 %):
#   TG_NS_NT (non-temporal output stores), TG_NS_THREADS (workgroup size), --idx32 / --ptr32 (u32 shadows)
for cfg in "1 512 1 1" "0 512 1 1" "1 256 1 1" "1 1024 1 1" "1 512 0 0" "1 512 1 1"; do
  set -- $cfg
  r=$(TG_NS_NT=$1 TG_NS_THREADS=$2 python bench.py --no-cpu-baseline --idx32 $3 --ptr32 $4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['roofline']['avg_launch_ms'],4), round(d['value']/1e9,2))")
  echo "NT=$1 threads=$2 idx32=$3 ptr32=$4  ->  launch ms, G edges/s: $r"
done
