#!/usr/bin/env python3
"""resize(jbu_apply(src, kc)) vs jbu_apply_resized(src, jbu_blend(kc)): values and time."""
import sys, torch
sys.path.insert(0, ".")
from isegprobe_amd import hip_ops as ops
torch.manual_seed(0)
for (B, h, C) in ((2, 32, 128), (2, 16, 64), (32, 256, 384)):
    GH, OH = 2 * h, 2 * h * 7 // 8
    x = torch.randn(B, h, h, C, device="cuda").to(torch.bfloat16)
    kc = torch.rand(B, GH, GH, 8, 16, device="cuda") / 8
    xs = torch.arange(GH, device="cuda")
    bx = ((xs - 4) >> 1) - 1
    inwin = ((torch.arange(16, device="cuda")[None, :] - bx[:, None]) & 15) < 8
    kc = (kc * inwin[None, None, :, None, :]).to(torch.float16)
    def ref():
        return ops.resize_nhwc(ops.jbu_apply(x, kc), OH, OH, "bilinear")
    kc9 = ops.jbu_blend(kc, OH, OH)
    def new():
        return ops.jbu_apply_resized(x, kc9)
    r, n = ref().float(), new().float()
    d = (r - n).abs()
    out = [f"B={B} {h}->{GH}->{OH} C={C}: maxdiff {d.max().item():.3e} mean {d.mean().item():.2e} (ref rms {r.pow(2).mean().sqrt().item():.2e}) finite={bool(torch.isfinite(n).all())}"]
    for name, fn in (("apply+resize", ref), ("blend", lambda: ops.jbu_blend(kc, OH, OH)), ("apply_resized", new)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        out.append(f"{name} {s.elapsed_time(e)/5:.3f} ms")
    print(" | ".join(out))
