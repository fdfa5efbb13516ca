#!/usr/bin/env python3
"""jbu_kernels_resized vs jbu_blend(jbu_kernels): must agree bit for bit; times both."""
import sys, logging, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import seeded_
from isegprobe_amd import hip_ops as ops
from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
torch.manual_seed(0)
up = seeded_(JBUFeatUpUpsampler("dinov2"), 3).cuda().eval()
P = up.upsampler.up4.packed()
for (B, S, O) in ((2, 64, 56), (2, 72, 63), (3, 40, 35), (32, 512, 448)):
    g = torch.randn(B, 3, O, O, device="cuda")
    small = ops.adaptive_avg_pool(g, S, S)
    proj = ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"])
    two = lambda: ops.jbu_blend(ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"]), O, O)
    one = lambda: ops.jbu_kernels_resized(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"], O, O)
    a, b = two(), one()
    out = [f"B={B} {S}->{O}: equal={torch.equal(a, b)} maxdiff={(a.float()-b.float()).abs().max().item():.2e}"]
    for name, fn in (("kernels+blend", two), ("fused", one)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): fn()
        e.record(); torch.cuda.synchronize()
        out.append(f"{name} {s.elapsed_time(e)/5:.3f} ms")
    print(" | ".join(out))
