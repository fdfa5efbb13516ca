#!/usr/bin/env python3
"""Dense GEMM latency at the batch-2 click-loop shapes (M = 2050 tokens)."""
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
for (M, K, N) in ((2050, 384, 1152), (2050, 384, 384), (2050, 384, 1536), (2050, 1536, 384), (514, 384, 1152)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") / 20).to(torch.bfloat16); b = torch.randn(N, device="cuda")
    y = ops.linear(A, W, b); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): y = ops.linear(A, W, b)
    e.record(); torch.cuda.synchronize()
    ref = A.float() @ W.float().t() + b
    print(f"M={M} K={K} N={N}: {s.elapsed_time(e)/50*1e3:.1f} us  relerr {(y.float()-ref).abs().max().item()/ref.abs().max().item():.1e}")
