#!/usr/bin/env python3
"""Condenses rocprofv3 csv output (kernel stats + PMC passes) into a short text summary."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        print({k: row[k] for k in row if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
KERNELS = ("screen_kernel", "exact_kernel", "full_kernel", "validity_kernel")
print("== PMC (per dispatch averages, per kernel) ==")
for f in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    acc = defaultdict(list)
    for row in csv.DictReader(open(f)):
        kn = next((k for k in KERNELS if k in row.get("Kernel_Name", "")), None)
        if kn:
            acc[(kn, row["Counter_Name"])].append(float(row["Counter_Value"]))
    for (kn, k), v in sorted(acc.items()):
        print(f"{kn} {k}: mean {sum(v)/len(v):.6g} over {len(v)} dispatches")
