#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <csv> [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:118]:118s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f}us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
