#!/usr/bin/env python3
"""The blend kernel alone at the configs[0] shape (batch 32, 32 x 32 -> 448 x 448, 384 channels), 5 launches: workload of PMC passes."""
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B, h, w, H, W, N = 32, 32, 32, 448, 448, 384
if len(sys.argv) > 1:
    B, h, w, H, W, N = (int(v) for v in sys.argv[1:7])
z = (torch.randn(B * h * w, 9 * N, device="cuda") * 0.3).half()
bias = torch.randn(N, device="cuda") * 0.1
for _ in range(5):
    y = ops.conv3x3_of_bilinear_blend(z, bias, B, h, w, H, W, N)
torch.cuda.synchronize()
print("ok", float(y.float().mean()))
import os
def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n
t = timed(lambda: ops.conv3x3_of_bilinear_blend(z, bias, B, h, w, H, W, N))
print(f"blend (form {os.environ.get('ISEGPROBE_BLEND_FORM', '4')}, ablation builds: -DISP_BLEND_ABLATE via ISEGPROBE_HIP_LIB): {t:.3f} ms")
xb = torch.randn(B, h, w, N, device="cuda").to(torch.bfloat16)
t = timed(lambda: ops.resize_nhwc(xb, H, W, "bilinear"))
print(f"plain bilinear resize kernel writing the same {B * H * W * N * 2 / 1e9:.2f} GB map: {t:.3f} ms")
