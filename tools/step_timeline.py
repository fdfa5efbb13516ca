#!/usr/bin/env python3
"""Timeline of ONE headline step from a rocprofv3 --kernel-trace CSV: per queue, when its kernels start / end relative to the
step's first kernel, and the idle gaps of the main queue.  usage: step_timeline.py <kernel_trace.csv> [step index from the end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
starts = [i for i, r in enumerate(rows) if "click_maps_kernel" in r["Kernel_Name"]]
i0 = starts[-back - 1] if len(starts) > back else starts[0]
i1 = starts[-back] if back > 0 and len(starts) > back else len(rows)
step = rows[i0:i1]
t0 = int(step[0]["Start_Timestamp"])
queues = {}
for r in step:
    queues.setdefault(r["Queue_Id"], []).append(r)
print(f"step: {len(step)} kernels, {(max(int(r['End_Timestamp']) for r in step) - t0) / 1e6:.3f} ms, queues {list(queues)}")
for q, rs in queues.items():
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
    print(f"-- queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, from {(int(rs[0]['Start_Timestamp']) - t0) / 1e6:.3f} to {(int(rs[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
    prev_end = None
    groups = []
    for r in rs:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        gap = 0 if prev_end is None else s - prev_end
        if groups and groups[-1][0] == name(r) and gap < 20000:
            groups[-1][2] = e
            groups[-1][3] += 1
            groups[-1][4] += e - s
        else:
            groups.append([name(r), s, e, 1, e - s, gap])
        prev_end = e
    for g in groups:
        if g[4] > 150000 or g[5] > 50000:
            print(f"   {g[1] / 1e6:8.3f} .. {g[2] / 1e6:8.3f} ms  x{g[3]:<3d} busy {g[4] / 1e6:7.3f}  gap before {g[5] / 1e3:7.1f} us  {g[0]}")
