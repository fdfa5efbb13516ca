#!/usr/bin/env python3
"""ViT-shaped dense GEMMs: this repo's gemm_tile_kernel against the vendor library torch dispatches to (hipBLASLt /
rocBLAS), same operands, HIP-event time.  A measurement only: the product path does not link the vendor library."""
import sys
import torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 32800


def timed(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, K, N, act in (("qkv", 384, 1152, None), ("fc1+gelu", 384, 1536, "gelu"), ("proj", 384, 384, None), ("fc2", 1536, 384, None)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    bb = b.to(torch.bfloat16)
    t_own = timed(lambda: ops.linear(a, w, b, act))
    if act == "gelu":
        t_lib = timed(lambda: F.gelu(F.linear(a, w, bb)))
        t_lib_mm = timed(lambda: F.linear(a, w, bb))
    else:
        t_lib = t_lib_mm = timed(lambda: F.linear(a, w, bb))
    fl = 2.0 * M * K * N
    print(f"{name:9s} M={M} K={K} N={N}: own {t_own:6.1f} us ({fl / t_own / 1e6:5.0f} TF/s)   library {t_lib:6.1f} us "
          f"(GEMM+bias alone {t_lib_mm:6.1f} us, {fl / t_lib_mm / 1e6:5.0f} TF/s)")
