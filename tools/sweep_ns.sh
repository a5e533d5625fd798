#!/bin/bash
# A/B sweep of the headline kernel's tuning knobs on ONE box (boxes differ by ~10 %):
#   TG_NS_NT (non-temporal output stores), TG_NS_NTLOAD (streaming gathers), TG_NS_THREADS, --idx32 / --ptr32
for cfg in "1 1 512 1 1" "0 1 512 1 1" "1 0 512 1 1" "1 1 256 1 1" "1 1 512 0 0" "1 1 512 1 1"; do
  set -- $cfg
  r=$(TG_NS_NT=$1 TG_NS_NTLOAD=$2 TG_NS_THREADS=$3 python bench.py --steps 8192 --warmup 1024 --no-cpu-baseline --idx32 $4 --ptr32 $5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['roofline']['avg_launch_ms'],4), round(d['value']/1e9,2))")
  echo "NT=$1 NTLOAD=$2 threads=$3 idx32=$4 ptr32=$5  ->  launch ms, G edges/s: $r"
done
