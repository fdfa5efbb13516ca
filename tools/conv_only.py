#!/usr/bin/env python3
"""Run only the head conv a few times (for PMC passes)."""
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
x = torch.randn(B, 448, 448, 384, device="cuda").to(torch.bfloat16)
w = (torch.randn(384, 9 * 384, device="cuda") / 60).to(torch.bfloat16)
bias = torch.randn(384, device="cuda")
for _ in range(4):
    y = ops.conv3x3(x, w, bias, "relu")
torch.cuda.synchronize()
