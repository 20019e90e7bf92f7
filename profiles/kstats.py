#!/usr/bin/env python3
"""print per-kernel durations from rocprofv3 kernel_stats csv files under a directory"""
import csv, glob, sys
for d in sys.argv[1:]:
    print("==", d)
    for f in sorted(set(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))):
        for r in csv.DictReader(open(f)):
            n = r["Name"]
            if "mvk" in n:
                short = n.split("namespace)::")[-1].split("(")[0]
                print(f"  {short:40s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']}%")
