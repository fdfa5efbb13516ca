#!/usr/bin/env python3
"""The ViT's dense GEMM shapes (M = 32 x 1025 tokens, IEEE-half operands) alone, for A/B builds of csrc/gemm.hip selected
through ISEGPROBE_HIP_LIB (-DISP_ABLATE_NO_DMA: no in-loop LDS-DMA; -DISP_ABLATE_NO_MFMA: DMA + barriers only;
-DISP_ABLATE_GEMM_NO_EPILOGUE) -- which part of a 128 x 128 x 384 tile's ~9.5 us is staging latency, MFMA work, epilogue."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from isegprobe_amd import hip_ops as ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32800
F16 = torch.float16
def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for name, K, N, act in (("qkv", 384, 1152, None), ("fc1", 384, 1536, "gelu"), ("proj", 384, 384, "res"), ("fc2", 1536, 384, "res")):
    A = torch.randn(M, K, device="cuda").to(F16)
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(F16)
    b = torch.randn(N, device="cuda") * 0.1
    if act == "res":
        x = torch.randn(M, N, device="cuda")
        g = torch.ones(N, device="cuda")
        us = timed(lambda: ops.linear_residual_(x, A, W, b, g))
    else:
        us = timed(lambda: ops.linear(A, W, b, act))
    out.append(f"{name} {us:.1f} us ({2.0 * M * K * N / us / 1e6:.0f} TFLOP/s)")
print(os.environ.get("ISEGPROBE_HIP_LIB", "shipped").split("/")[-1] + ": " + "  ".join(out))
