#!/usr/bin/env python3
"""Summarise tools/bench_pmc.sh: per-kernel launch durations (kernel-stats) and fabric traffic (PMC passes) of bench.py's
own launches -> profiles/<round>_bench_jbu448_b32_kernel_stats.csv, <round>_bench_pmc.json, <round>_conv_pmc.json (ROUND env, default r04)."""
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1] if len(sys.argv) > 1 else "profiles"
R = os.environ.get("ROUND", "r04")


def latest(pattern):
    files = sorted(glob.glob(pattern, recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n


ks = latest(f"gpurun_out/{R}_kstats/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(out, f"{R}_bench_jbu448_b32_kernel_stats.csv"))
per = {}
for d, counters in ((f"{R}_pmc_fetch", ("FETCH_SIZE",)), (f"{R}_pmc_write", ("WRITE_SIZE",)), (f"{R}_pmc_l2", ("TCC_HIT_sum", "TCC_MISS_sum"))):
    f = latest(f"gpurun_out/{d}/**/*counter_collection.csv")
    if not f:
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        k = short(row["Kernel_Name"])
        acc.setdefault(k, {}).setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
        acc[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k, cs in acc.items():
        for c, disp in cs.items():
            vals = list(disp.values())
            per.setdefault(k, {})[c] = {"per_launch_mean": sum(vals) / len(vals), "launches": len(vals)}
res = {"command": "rocprofv3 --pmc <counter> --output-format csv -- python bench.py --steps 3 --warmup 1 --no-stages --no-cpu-baseline "
                  "(one pass per counter group, tools/bench_pmc.sh)",
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE reports 1/2 of a wide coalesced read stream "
                "(MI355X_MICROARCH.md, HBM) -> bytes = FETCH_SIZE*1024*2 + WRITE_SIZE*1024; the counters sit on the L2's fabric side "
                "(Infinity-Cache hits included): an upper bound on HBM bytes",
       "kernels": {}}
for k, cs in sorted(per.items()):
    e = {c: v for c, v in cs.items()}
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        e["traffic_bytes_per_launch"] = cs["FETCH_SIZE"]["per_launch_mean"] * 1024 * 2 + cs["WRITE_SIZE"]["per_launch_mean"] * 1024
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        h, m = cs["TCC_HIT_sum"]["per_launch_mean"], cs["TCC_MISS_sum"]["per_launch_mean"]
        e["l2_hit_rate"] = h / (h + m) if h + m else None
    res["kernels"][k] = e
json.dump(res, open(os.path.join(out, f"{R}_bench_pmc.json"), "w"), indent=1)
conv = {k: v for k, v in res["kernels"].items() if "conv3x3_patch4_kernel_192" in k and "traffic_bytes_per_launch" in v}
if conv:
    mean = sum(v["traffic_bytes_per_launch"] for v in conv.values()) / len(conv)
    json.dump({"command": res["command"], "kernels": conv,
               "derived": {"traffic_bytes_per_launch": mean, "algorithmic_bytes_per_launch": 9867657216,
                           "note": "mean over the two head-conv launches of a bench step (folded-affine first conv, classifier-fused "
                                   "second conv); " + res["units"]}},
              open(os.path.join(out, f"{R}_conv_pmc.json"), "w"), indent=1)
    print("conv traffic per launch:", {k[-60:]: round(v["traffic_bytes_per_launch"] / 1e9, 2) for k, v in conv.items()})
jbu = {k: v for k, v in res["kernels"].items() if k.startswith(("jbu_", "adaptive_avg_pool")) and "traffic_bytes_per_launch" in v}
print("JBU kernels GB per launch x launches:", {k[:40]: (round(v["traffic_bytes_per_launch"] / 1e9, 3), v["FETCH_SIZE"]["launches"]) for k, v in jbu.items()})
