"""rocprofv3 --pmc CSVs -> per kernel (substring match) and counter: min / max / mean over its dispatches
(the windowed sampler launches every kernel once per hop: min ~ hop 1, max ~ the last hop)."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, needles = sys.argv[1], sys.argv[2:] or [""]
vals = defaultdict(lambda: defaultdict(float))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        for nd in needles:
            if nd in row["Kernel_Name"]:
                vals[(nd, row["Counter_Name"])][row["Dispatch_Id"]] += float(row["Counter_Value"])
out = {}
for (nd, c), d in sorted(vals.items()):
    v = sorted(d.values())
    out["%s:%s" % (nd, c)] = {"min": v[0], "max": v[-1], "mean": sum(v) / len(v), "dispatches": len(v)}
print(json.dumps(out, indent=1))
