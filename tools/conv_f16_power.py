#!/usr/bin/env python3
"""Data-dependent speed of the half-precision head conv (the socket is power-limited during it): full half weights vs half
weights that hold bf16-rounded values (3 trailing zero mantissa bits) vs the bf16 kernel.  usage: conv_f16_power.py [B]"""
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
x = torch.randn(B, 448, 448, 384, device="cuda")
w = torch.randn(384, 9 * 384, device="cuda") / 60
bias = torch.randn(384, device="cuda")


def timed(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


xh, xb = x.half(), x.to(torch.bfloat16)
cases = {"f16 x, f16 w": (xh, w.half()), "f16 x, w = bf16-rounded values in f16": (xh, w.to(torch.bfloat16).half()),
         "x, w = bf16-rounded values in f16": (xb.half(), w.to(torch.bfloat16).half()), "bf16 kernel": (xb, w.to(torch.bfloat16))}
for rep in range(2):
    for name, (xx, ww) in cases.items():
        ms = timed(lambda: ops.conv3x3(xx, ww, bias, "relu"))
        print(f"{name:44s} {ms:.3f} ms  {2.0 * B * 448 * 448 * 384 * 9 * 384 / ms / 1e9:.0f} TFLOP/s", flush=True)
