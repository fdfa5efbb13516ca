#!/usr/bin/env python3
"""BASELINE configs[0]'s model on the GPU -- DINOv2-S/14 + bilinear + ConvSegHead(384,2,1), batch 32 at 448^2, forward only -- for
rocprofv3 --kernel-trace --stats (per-kernel split of bench.py's cfg0_bilinear448 block).  usage: cfg0_only.py [B] [S] [iters]"""
import logging
import os
import sys
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
logging.getLogger("root").setLevel(logging.WARNING)
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 448
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
model = bench.build("bilinear", S, "dinov2_vits14").cuda()
image, points = bench.synthetic_batch(B, S, 448)
image, points = image.cuda(), points.cuda()
dt, out = bench._time_forward(model, image, points, 2, iters)
print(f"S/14 + bilinear + ConvSegHead B={B} {S}x{S}: {dt * 1e3:.2f} ms/step = {B / dt:.0f} img/s")
