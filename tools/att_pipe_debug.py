#!/usr/bin/env python3
"""Localise errors of the pipelined attention kernel: per stream / d-block / key-structure error maps."""
import math, sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

def ref(q, k, v):
    sc = q.double().permute(0, 2, 1, 3) @ k.double().permute(0, 2, 3, 1) * math.log(2.0)
    return (sc.softmax(-1) @ v.double().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)

torch.manual_seed(0)
for hd in (64, 128):
    for Lk in (128, 192, 256, 1024):
        B, H, Lq = 1, 1, 256
        q = (torch.randn(B, Lq, H, hd, device="cuda") * (hd ** -0.5 * 1.44)).to(torch.bfloat16)
        k = torch.randn(B, Lk, H, hd, device="cuda").to(torch.bfloat16)
        v = torch.randn(B, Lk, H, hd, device="cuda").to(torch.bfloat16)
        out = ops.attention(q, k, v, None, q_logit2=True).double()
        e = (out - ref(q, k, v)).abs()[0, :, 0]          # [Lq, hd]
        print(f"hd {hd} Lk {Lk}: max {e.max():.4f}; per wave/stream (32-query groups): {[round(x, 3) for x in e.view(8, 32, hd).amax((1, 2)).tolist()]}; "
              f"per 32-wide d block: {[round(x, 3) for x in e.view(Lq, hd // 32, 32).amax((0, 2)).tolist()]}")
        if e.max() > 0.02:
            # which keys matter: one-hot V probes -> out[q, d] = sum_k P[q,k] V[k,d]; with V[k,:] = onehot(k % hd) we see P folded mod hd
            v2 = torch.zeros_like(v)
            idx = torch.arange(Lk, device="cuda")
            v2[0, idx, 0, idx % hd] = 1.0
            o2 = ops.attention(q, k, v2, None, q_logit2=True).double()
            r2 = ref(q, k, v2)
            d2 = (o2 - r2).abs()[0, :, 0]
            print("   one-hot V probe: max err per d (first 32):", [round(x, 3) for x in d2.amax(0)[:32].tolist()])
            # uniform V: checks the normaliser only
            v3 = torch.ones_like(v)
            o3 = ops.attention(q, k, v3, None, q_logit2=True).double()
            print("   all-ones V (should be 1): min %.4f max %.4f" % (o3.min().item(), o3.max().item()))
