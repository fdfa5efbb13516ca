#!/usr/bin/env python3
"""Time the head conv (448^2, C=N=384) with HIP events; run once per ISEGPROBE_CONV_ENGINE setting."""
import os
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
relu_in = len(sys.argv) > 2 and sys.argv[2] == "relu"
x = torch.randn(B, 448, 448, 384, device="cuda")
if relu_in:
    x = x.relu()
x = x.to(torch.bfloat16)
w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
bias = torch.randn(384, device="cuda")
for _ in range(3):
    y = ops.conv3x3(x, w, bias, "relu")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 10
e0.record()
for _ in range(n):
    y = ops.conv3x3(x, w, bias, "relu")
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * B * 448 * 448 * 384 * 9 * 384
print(f"engine={os.environ.get('ISEGPROBE_CONV_ENGINE', 'patch')} B={B} relu_in={relu_in} {ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
