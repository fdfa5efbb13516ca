#!/usr/bin/env python3
"""Per-stage timing of the FeatUp-JBU stack at the bench shape (B=32, 448^2 guidance, 32x32x384 source)."""
import sys, logging
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import seeded_
from isegprobe_amd import hip_ops as ops
from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
up = seeded_(JBUFeatUpUpsampler("dinov2"), 3).cuda().eval()
x = torch.randn(B, 32, 32, 384, device="cuda").to(torch.bfloat16)
g = torch.randn(B, 3, 448, 448, device="cuda")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): r = fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n, r
tot = 0
with torch.no_grad():
    for i, st in enumerate((up.upsampler.up1, up.upsampler.up2, up.upsampler.up3, up.upsampler.up4)):
        P = st.packed()
        GH = x.shape[1] * 2
        tp, small = t(lambda: ops.adaptive_avg_pool(g, GH, GH))
        tr, proj = t(lambda: ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"]))
        tk, kc = t(lambda: ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"]))
        ta, y = t(lambda: ops.jbu_apply(x, kc))
        px = B * GH * GH
        print(f"stage {i+1} -> {GH}^2: pool {tp:.3f}  proj {tr:.3f}  kernels {tk:.3f} ({px*(128+256+12)/tk/1e6:.0f} GB/s)  "
              f"apply {ta:.3f} ms ({px*64*384*2/ta/1e9:.0f} TFLOP/s dense-equivalent, {(px*384*2*1.25+px*256)/ta/1e6:.0f} GB/s)")
        tot += tp + tr + tk + ta
        x = y
print(f"total {tot:.2f} ms")
