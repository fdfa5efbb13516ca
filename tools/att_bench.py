#!/usr/bin/env python3
"""ViT self-attention alone (packed qkv, head_dim 64): HIP-event time per launch, both entry points.
usage: att_bench.py [B] [L] [heads] [iters]   (ISEGPROBE_ATT64=0 selects the generic kernel)"""
import sys
import torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1025
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 50
torch.manual_seed(0)
qkv = torch.randn(B * L, 3 * heads * 64, device="cuda").to(torch.bfloat16)
flops = 4.0 * B * heads * L * L * 64


def timed(fn):
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


for name, fn in (("scale argument", lambda: ops.attention_packed_qkv(qkv, B, L, heads, 0.125)),
                 ("q carries scale*log2e", lambda: ops.attention_packed_qkv(qkv, B, L, heads, None, q_logit2=True))):
    us = timed(fn)
    print(f"B={B} L={L} heads={heads} {name}: {us:.1f} us  {flops / us / 1e6:.0f} TFLOP/s")
