#!/usr/bin/env python3
"""Summarise the per-counter rocprofv3 CSVs that tools/conv_pmc.sh wrote into profiles/<out>.json."""
import csv, glob, json, os, sys
out = sys.argv[1]
kernel_sub = sys.argv[2] if len(sys.argv) > 2 else "conv3x3_patch"
res = {}
for d in ("pmc_fetch", "pmc_write", "pmc_l2"):
    files = sorted(glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not files:
        continue
    per = {}
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if kernel_sub not in row["Kernel_Name"]:
                continue
            per.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for name, disp in per.items():
        vals = list(disp.values())
        res[name] = {"per_launch_mean": sum(vals) / len(vals), "launches": len(vals)}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
