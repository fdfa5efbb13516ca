#!/usr/bin/env python3
"""Residual-epilogue GEMMs alone (HIP events, 20 launches): LoftUp's [1.6 M x 512] x [512 -> 448] with residual + row
statistics, the ViT's proj / fc2 with the fp32 residual stream.  A/B builds through ISEGPROBE_HIP_LIB."""
import os, sys, math
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT)
from isegprobe_amd import hip_ops as ops


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


H = torch.float16
M = 8 * 448 * 448
a = torch.randn(M, 512, device="cuda", dtype=H)
w = (torch.randn(448, 512, device="cuda") / math.sqrt(512)).to(H)
b = torch.randn(448, device="cuda")
res = torch.randn(M, 448, device="cuda", dtype=H)
print(f"loftup wo  (res+stats) M={M} K=512 N=448: {timeit(lambda: ops.linear_axpy_res_stats(a, w, b, res, 1.0)):8.1f} us")
print(f"loftup wo  (res)       M={M} K=512 N=448: {timeit(lambda: ops.linear_axpy_res(a, w, b, res, 1.0)):8.1f} us")
print(f"loftup     (bias only) M={M} K=512 N=448: {timeit(lambda: ops.linear(a, w, b)):8.1f} us")
del a, res
Mv = 32 * 1025
x = torch.randn(Mv, 384, device="cuda")
for K in (384, 1536):
    av = torch.randn(Mv, K, device="cuda", dtype=H)
    wv = (torch.randn(384, K, device="cuda") / math.sqrt(K)).to(H)
    g = torch.rand(384, device="cuda")
    print(f"vit residual fp32 stream M={Mv} K={K} N=384: {timeit(lambda: ops.linear_residual_(x, av, wv, b[:384].contiguous(), g), 50):8.1f} us")
af = torch.randn(Mv, 384, device="cuda", dtype=H)
wf = (torch.randn(1536, 384, device="cuda") / math.sqrt(384)).to(H)
bf = torch.randn(1536, device="cuda")
print(f"vit fc1 + GELU (half)    M={Mv} K=384 N=1536: {timeit(lambda: ops.linear(af, wf, bf, 'gelu'), 50):8.1f} us")
print(f"vit fc1 bias only (half) M={Mv} K=384 N=1536: {timeit(lambda: ops.linear(af, wf, bf), 50):8.1f} us")
