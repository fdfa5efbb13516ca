#!/usr/bin/env python3
"""On-box roofline probes: dense bf16 MFMA rate (register-resident loop, zeros vs random operands) and HBM copy rate."""
import ctypes
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import _lib
L = _lib.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def ev():
    return torch.cuda.Event(enable_timing=True)
sink = torch.zeros(1 << 20, device="cuda")
for name, seed in (("zeros", torch.zeros(65536, device="cuda").to(torch.bfloat16)),
                   ("random", torch.randn(65536, device="cuda").to(torch.bfloat16))):
    blocks, iters = 256 * 8, 20000
    for _ in range(2):
        L.isp_probe_mfma_bf16(seed.data_ptr(), sink.data_ptr(), blocks, iters, st())
    torch.cuda.synchronize()
    s, e = ev(), ev()
    s.record()
    for _ in range(3):
        L.isp_probe_mfma_bf16(seed.data_ptr(), sink.data_ptr(), blocks, iters, st())
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    fl = blocks * 4 * iters * 16 * 2.0 * 16 * 16 * 32
    print(f"MFMA bf16 16x16x32, {name:6s} operands: {fl / ms / 1e9:7.0f} TFLOP/s  ({ms:.1f} ms)")
    for _ in range(2):
        L.isp_probe_mfma_bf16_32x32(seed.data_ptr(), sink.data_ptr(), blocks, iters, st())
    torch.cuda.synchronize()
    s, e = ev(), ev()
    s.record()
    for _ in range(3):
        L.isp_probe_mfma_bf16_32x32(seed.data_ptr(), sink.data_ptr(), blocks, iters, st())
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 3
    fl = blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16
    print(f"MFMA bf16 32x32x16, {name:6s} operands: {fl / ms / 1e9:7.0f} TFLOP/s  ({ms:.1f} ms)")
n = 4 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 255)
b = torch.empty_like(a)
for _ in range(2):
    L.isp_probe_copy(a.data_ptr(), b.data_ptr(), n, st())
torch.cuda.synchronize()
s, e = ev(), ev()
s.record()
for _ in range(5):
    L.isp_probe_copy(a.data_ptr(), b.data_ptr(), n, st())
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 5
print(f"float4 copy of 4 GiB: {2 * n / ms / 1e6:.0f} GB/s read+write  ({ms:.2f} ms)")
