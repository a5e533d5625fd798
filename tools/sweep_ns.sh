#!/bin/bash
# tuning sweep of the headline kernel's launch knobs (prints avg launch ms per setting)
for ntl in 0 1; do for th in 256 1024; do
  r=$(TG_NS_NTLOAD=$ntl TG_NS_THREADS=$th python bench.py --steps 4096 --warmup 1024 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['roofline']['avg_launch_ms'],4), round(d['value']/1e9,2))")
  echo "NTLOAD=$ntl threads=$th  ->  $r"
done; done
