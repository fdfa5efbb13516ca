#!/usr/bin/env python3
"""Workload of tools/mlp_pmc.sh: the ViT block's MLP branch at the headline shape (32 images x 1025 tokens, D 384, hidden 1536, IEEE
half) -- five launches of the fused per-image-tiled kernel (patch rows) and five of the three-kernel route (all rows)."""
import math
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

D, HID, B, T = 384, 1536, 32, 1024
torch.manual_seed(0)
H = torch.float16
nw, nb = torch.randn(D, device="cuda") * 0.3 + 1, torch.randn(D, device="cuda") * 0.2
w1, b1 = torch.randn(HID, D, device="cuda") / math.sqrt(D), torch.randn(HID, device="cuda") * 0.3
w2, b2 = torch.randn(D, HID, device="cuda") / math.sqrt(HID), torch.randn(D, device="cuda") * 0.3
ls = torch.randn(D, device="cuda") * 0.5 + 1
P = ops.vit_mlp_pack(nw, nb, w1, b1, w2, b2, ls, dtype=H)
w1h, w2h = w1.to(H), w2.to(H)
x = torch.randn(B * (T + 1), D, device="cuda")
for _ in range(5):
    ops.vit_mlp_fused_rows_(x, *P, 1e-6, B, T + 1, 1, T)
for _ in range(5):
    h = ops.layernorm(x, nw, nb, 1e-6, out_dtype=H)
    hid = ops.linear(h, w1h, b1, "gelu")
    ops.linear_residual_(x, hid, w2h, b2, ls)
torch.cuda.synchronize()
print("done")
