#!/usr/bin/env python3
"""Dense GEMM throughput for the short-K shapes of the path (LoftUp / ViT)."""
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
def run(M, K, N, act=None):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    W = (torch.randn(N, K, device="cuda") / 20).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    y = ops.linear(A, W, b, act); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): y = ops.linear(A, W, b, act)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"M={M} K={K} N={N}: {ms:.3f} ms  {2.0*M*K*N/ms/1e9:.0f} TFLOP/s  io {(M*K+M*N)*2/ms/1e6:.0f} GB/s")
for shp in ((1605632, 448, 512), (1605632, 512, 448), (1605632, 448, 384), (1605632, 384, 448), (32800, 384, 1152), (32800, 384, 1536), (32800, 1536, 384), (401408, 384, 384), (1605632, 384, 384)):
    run(*shp)
