#!/usr/bin/env python3
import sys, torch, numpy as np
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
for B, S in ((8, 224), (8, 448), (32, 448)):
    gt = torch.zeros(B, 1, S, S, device="cuda"); gt[:, :, S // 4: 3 * S // 4, S // 5: S // 2] = 1
    pred = torch.rand(B, 1, S, S, device="cuda")
    pts = -torch.ones(B, 48, 3, device="cuda")
    draws = torch.randint(0, 2 ** 32, (B,), dtype=torch.int64)
    out, ws = ops.next_points(pred, gt, pts, 1, draws)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): ops.next_points(pred, gt, pts, 1, draws, workspace=ws)
    e.record(); torch.cuda.synchronize()
    print(f"next_points B={B} {S}x{S}: {s.elapsed_time(e) / 5:.2f} ms")
