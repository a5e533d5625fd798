"""Timeline of the kernels of ONE launch from a rocprofv3 --kernel-trace CSV: start / end offsets (us) per kernel, so
that overlap between streams can be read off.  usage: python tools/timeline.py <dir with *kernel_trace.csv> [launch index]"""
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
starts = [i for i, r in enumerate(rows) if "win_vtab_kernel" in r[2] or "win_init_kernel" in r[2] or "win_first_hops" in r[2]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) - 2
lo = starts[which]
hi = starts[which + 1] if which + 1 < len(starts) else len(rows)
t0 = rows[lo][0]
for s, e, name, q in rows[lo:hi]:
    short = name.split("(")[0].replace("void tg::", "")[:60]
    print("%9.1f %9.1f  %8.1f us  q%s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, short))
print("launch span %.1f us" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3))
