#!/bin/bash
# SQ-side counters of the windowed launch's kernels (what the wavefronts of K1 / K4 spend their cycles on)
# -> gpurun_out/pmc_sq_<tag>.txt.  One --pmc pass per group of counters, kernel-trace not combined with anything else.
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
ARGS="--form windowed --no-secondary --no-cpu-baseline --steps 2 --warmup 1 $@"
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT" "GRBM_GUI_ACTIVE GRBM_COUNT TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $ROOT/gpurun_out/pmc_sq_${tag}_$i --output-format csv -- python3 $ROOT/bench.py $ARGS > $ROOT/gpurun_out/pmc_sq_${tag}_$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $ROOT
python3 - <<PY > gpurun_out/pmc_sq_$tag.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/pmc_sq_${tag}_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            for key in ("win_stage_gather", "win_stage_emit", "win_stage_first", "win_sort_fine", "win_scatter8",
                        "win_gather", "win_emit", "win_scatter", "win_hist"):
                if key in name:
                    direct = "DIRECT" if ("true>" in name.replace(" ", "") and key == "win_emit") else ""
                    acc[key + direct][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    break
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-32s n=%d mean %.4g min %.4g max %.4g" % (c, len(v), sum(v) / len(v), min(v), max(v)))
PY
cat gpurun_out/pmc_sq_$tag.txt
