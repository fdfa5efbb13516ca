#!/usr/bin/env python3
"""Per-kernel timing on the GPU box (HIP events on the current stream)."""
import math
import sys

import torch

sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = "cuda"
    # head conv at 448^2, C=384
    x = torch.randn(B, 448, 448, 384, device=dev).to(BF)
    w = (torch.randn(384, 9 * 384, device=dev) / 60).to(BF)
    bias = torch.randn(384, device=dev)
    ms = timeit(lambda: ops.conv3x3(x, w, bias, "relu"), 5, 2)
    fl = 2 * B * 448 * 448 * 384 * 9 * 384
    print(f"conv3x3 B={B} 448^2 C=384: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
    del x
    # ViT GEMMs at M = B4*1025 (B4 = 4B images)
    Bv = 32
    M = Bv * 1025
    for N, K, name in ((1152, 384, "qkv"), (384, 384, "proj"), (1536, 384, "fc1"), (384, 1536, "fc2")):
        A = torch.randn(M, K, device=dev).to(BF)
        W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(BF)
        bb = torch.randn(N, device=dev)
        ms = timeit(lambda: ops.linear(A, W, bb, None))
        print(f"gemm {name} M={M} N={N} K={K}: {ms:.3f} ms  {2 * M * N * K / ms / 1e9:.1f} TFLOP/s")
    qkv = torch.randn(M, 1152, device=dev).to(BF)
    ms = timeit(lambda: ops.attention_packed_qkv(qkv, Bv, 1025, 6, 0.125))
    fl = 4 * Bv * 6 * 1025 * 1025 * 64
    print(f"attention B={Bv} L=1025 h=6: {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s")
    xx = torch.randn(M, 384, device=dev)
    g = torch.randn(384, device=dev)
    ms = timeit(lambda: ops.layernorm(xx, g, g, 1e-6))
    print(f"layernorm rows={M} D=384: {ms:.3f} ms  {M * 384 * 6 / ms / 1e6:.1f} GB/s")
    f = torch.randn(B, 32, 32, 384, device=dev).to(BF)
    ms = timeit(lambda: ops.resize_bilinear_nhwc(f, 448, 448))
    print(f"bilinear B={B} 32->448 C=384: {ms:.3f} ms  {B * 448 * 448 * 384 * 2 / ms / 1e6:.1f} GB/s (write)")
    y = torch.randn(B, 448, 448, 384, device=dev).to(BF)
    ms = timeit(lambda: ops.classifier(y, g, 0.1))
    print(f"classifier B={B}: {ms:.3f} ms  {B * 448 * 448 * 384 * 2 / ms / 1e6:.1f} GB/s (read)")
    pts = torch.full((32, 48, 3), -1.0, device=dev)
    pts[:, :10, :2] = torch.randint(0, 448, (32, 10, 2), device=dev).float()
    pts[:, 24:30, :2] = torch.randint(0, 448, (32, 6, 2), device=dev).float()
    ms = timeit(lambda: ops.click_maps(pts, 448, 448, 5, 1.0, True))
    print(f"click_maps B=32 P=24 448^2: {ms:.3f} ms  {32 * 2 * 448 * 448 * 4 / ms / 1e6:.1f} GB/s")


if __name__ == "__main__":
    main()
