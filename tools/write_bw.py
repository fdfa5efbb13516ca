#!/usr/bin/env python3
"""On-box ceilings for a WRITE-dominated kernel (the blend of the through-the-resize convolution writes 4.93 GB and reads 0.23):
a 16-bit fill of the same size, a copy (read + write), and the library's own float4 copy probe."""
import sys
import torch
sys.path.insert(0, ".")


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


n = 32 * 448 * 448 * 384
x = torch.empty(n, device="cuda", dtype=torch.float16)
y = torch.empty_like(x)
t = timed(lambda: x.fill_(1.0))
print(f"fill   {n * 2 / 1e9:.2f} GB: {t:.3f} ms = {n * 2 / t / 1e6:.0f} GB/s written")
t = timed(lambda: x.zero_())
print(f"zero   {n * 2 / 1e9:.2f} GB: {t:.3f} ms = {n * 2 / t / 1e6:.0f} GB/s written")
t = timed(lambda: y.copy_(x))
print(f"copy   {n * 2 / 1e9:.2f} GB: {t:.3f} ms = {2 * n * 2 / t / 1e6:.0f} GB/s read + written")
xi = torch.empty(n, device="cuda", dtype=torch.int16)
t = timed(lambda: torch.arange(0, n, out=xi.view(torch.int16)) if False else xi.copy_(xi) if False else None)
a = torch.randn(n // 8, device="cuda", dtype=torch.float16)
t = timed(lambda: torch.mul(a, 1.5, out=x[: n // 8]))
print(f"mul    small warm {t:.3f} ms")
big = torch.empty(n, device="cuda", dtype=torch.float16)
src = torch.randn(4096, device="cuda", dtype=torch.float16)
t = timed(lambda: torch.mul(src.expand(n // 4096, 4096), 1.5, out=big.view(n // 4096, 4096)))
print(f"broadcast-mul (reads 8 KB, writes {n * 2 / 1e9:.2f} GB of non-constant data): {t:.3f} ms = {n * 2 / t / 1e6:.0f} GB/s written")
