#!/usr/bin/env python3
"""Fused ViT MLP kernel vs the LayerNorm + fc1(GELU) + fc2(residual) route.  usage: mlp_bench.py [M ...]"""
import math
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

D, HID = 384, 1536
torch.manual_seed(0)
nw, nb = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
w1, b1 = torch.randn(HID, D, device="cuda") / math.sqrt(D), torch.randn(HID, device="cuda") * 0.3
w2, b2 = torch.randn(D, HID, device="cuda") / math.sqrt(HID), torch.randn(D, device="cuda") * 0.3
ls = torch.randn(D, device="cuda") * 0.5 + 1
P = ops.vit_mlp_pack(nw, nb, w1, b1, w2, b2, ls)
w1b, w2b = w1.to(torch.bfloat16), w2.to(torch.bfloat16)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M in [int(a) for a in sys.argv[1:]] or [32768, 32800, 8200]:
    x = torch.randn(M, D, device="cuda")

    def old():
        h = ops.layernorm(x, nw, nb, 1e-6)
        hid = ops.linear(h, w1b, b1, "gelu")
        ops.linear_residual_(x, hid, w2b, b2, ls)
    fl = 4.0 * M * D * HID
    t_old = timed(old)
    x = torch.randn(M, D, device="cuda")
    t_new = timed(lambda: ops.vit_mlp_fused_(x, *P, 1e-6))
    print(f"M={M}: three kernels {t_old:.1f} us ({fl / t_old / 1e6:.0f} TFLOP/s)   fused {t_new:.1f} us ({fl / t_new / 1e6:.0f} TFLOP/s)")

# patch-token rows only of a [B, T + 1, D] stream (class-token rows left out): B x 1024 tokens = B x 8 exact tiles
B, T = 32, 1024
x = torch.randn(B * (T + 1), D, device="cuda")
t_rows = timed(lambda: ops.vit_mlp_fused_rows_(x, *P, 1e-6, B, T + 1, 1, T))
fl = 4.0 * B * T * D * HID
print(f"rows B={B} T={T}: fused {t_rows:.1f} us ({fl / t_rows / 1e6:.0f} TFLOP/s)")
# correctness of the row mapping and of the chunk rotation: against the contiguous kernel / the three-kernel route
x0 = torch.randn(B * (T + 1), D, device="cuda")
xa = x0.clone()
ops.vit_mlp_fused_rows_(xa, *P, 1e-6, B, T + 1, 1, T)
xb = x0.clone()
h = ops.layernorm(xb, nw, nb, 1e-6)
hid = ops.linear(h, w1b, b1, "gelu")
ops.linear_residual_(xb, hid, w2b, b2, ls)
v = xa.view(B, T + 1, D)
print("cls rows untouched:", torch.equal(v[:, 0], x0.view(B, T + 1, D)[:, 0]),
      " patch rows vs three kernels: max|d| %.3g (rms of the update %.3g)" % ((v[:, 1:] - xb.view(B, T + 1, D)[:, 1:]).abs().max().item(),
                                                                              (xb - x0).pow(2).mean().sqrt().item()))
