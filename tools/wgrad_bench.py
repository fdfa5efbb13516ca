#!/usr/bin/env python3
"""Head conv weight gradient at 448^2, C=N=384: fused nine-tap kernel vs nine pixel-reduction GEMMs."""
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H = W = 448
C = N = 384
g = torch.randn(B, H, W, N, device="cuda").to(torch.bfloat16)
x = torch.randn(B, H, W, C, device="cuda").relu().to(torch.bfloat16)
fl = 2.0 * B * H * W * 9 * C * N
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
def nine():
    dw = torch.zeros(N, 9 * C, device="cuda")
    M = B * H * W
    for t in range(9):
        ops.tn_gemm_atomic(g.view(M, N), x.view(M, C), dw[:, t * C:(t + 1) * C], shift=(H, W, t // 3 - 1, t % 3 - 1))
    return dw
a, b = timeit(lambda: ops.conv3x3_wgrad(g, x)), timeit(nine)
print(f"B={B}: fused {a:.2f} ms ({fl / a / 1e9:.0f} TFLOP/s)   nine GEMMs {b:.2f} ms ({fl / b / 1e9:.0f} TFLOP/s)   "
      f"rel diff {(ops.conv3x3_wgrad(g, x) - nine()).abs().max().item() / nine().abs().max().item():.2e}")
