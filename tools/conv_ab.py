#!/usr/bin/env python3
"""Time the head's first 3x3 conv (384 -> 384, 448^2) with HIP events: `conv_ab.py [batch] [iters]`.
Run once per setting of ISEGPROBE_CONV_PERSIST (read when the library first launches the kernel)."""
import os
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
x = torch.randn(B, 448, 448, 384, device="cuda").to(torch.bfloat16)
w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
bias = torch.randn(384, device="cuda")
for _ in range(3):
    y = ops.conv3x3(x, w, bias, "relu")
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(iters):
    y = ops.conv3x3(x, w, bias, "relu")
t1.record()
torch.cuda.synchronize()
ms = t0.elapsed_time(t1) / iters
flops = 2.0 * B * 448 * 448 * 384 * 9 * 384
print(f"persist={os.environ.get('ISEGPROBE_CONV_PERSIST', '1')} B={B}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
