"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_sum.py <dir> [kernel-substring] -> JSON
{counter: value per dispatch, averaged over the dispatches of the matching kernel}."""
import csv
import glob
import json
import sys
from collections import defaultdict

root, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc, calls = defaultdict(float), defaultdict(set)
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if needle in row["Kernel_Name"]:
            acc[row["Counter_Name"]] += float(row["Counter_Value"])
            calls[row["Counter_Name"]].add(row["Dispatch_Id"])
print(json.dumps({k: acc[k] / max(len(calls[k]), 1) for k in sorted(acc)}, indent=1))
