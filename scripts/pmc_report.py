"""Sum rocprofv3 --pmc counter CSVs per kernel: prints per-dispatch averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_trace"
tot = defaultdict(float)
disp = defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if pat not in name:
                continue
            c = row["Counter_Name"]
            tot[c] += float(row["Counter_Value"])
            disp[c].add((f, row.get("Dispatch_Id")))
print(f"kernel pattern: {pat}")
for c in sorted(tot):
    n = max(1, len(disp[c]))
    print(f"{c:28s} dispatches {n:4d}  total {tot[c]:.6g}  per-dispatch {tot[c] / n:.6g}")
