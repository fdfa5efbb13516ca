#!/usr/bin/env python3
"""LoftUp's cross-attention alone (batch 8, 448^2 queries x 1024 keys, 4 heads of 101 -> 128, IEEE half, base-2-logit
queries): HIP events over 10 launches.  ISEGPROBE_ATT128_NW=4 / ISEGPROBE_ATT128_DM=0 select the variants (default now: 4 waves; ISEGPROBE_ATT128_NW=8 the 8-wave form)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from isegprobe_amd import hip_ops as ops
B, Lq, Lk, H, hd = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 448 * 448, 1024, 4, 128
torch.manual_seed(0)
q = (torch.randn(B, Lq, H, hd, device="cuda") * 0.3).half()
k, v = torch.randn(B, Lk, H, hd, device="cuda").half(), torch.randn(B, Lk, H, hd, device="cuda").half()
for _ in range(2): o = ops.attention(q, k, v, None, q_logit2=True)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): o = ops.attention(q, k, v, None, q_logit2=True)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
fl = 4.0 * B * H * Lq * Lk * hd
print(f"attention hd128 B={B}: {ms:.3f} ms/launch, {fl / ms / 1e9:.0f} TFLOP/s executed ({fl * 101 / 128 / ms / 1e9:.0f} algorithmic at head_dim 101); "
      f"NW={os.environ.get('ISEGPROBE_ATT128_NW', '8')} DM={os.environ.get('ISEGPROBE_ATT128_DM', '1')}")
