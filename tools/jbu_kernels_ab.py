#!/usr/bin/env python3
"""Time jbu_kernels at the 512^2 / 256^2 stages and checksum its output (A/B of builds via ISEGPROBE_HIP_LIB)."""
import os, sys, logging
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import seeded_
from isegprobe_amd import hip_ops as ops
from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
B = 32
torch.manual_seed(0)
up = seeded_(JBUFeatUpUpsampler("dinov2"), 3).cuda().eval()
g = torch.randn(B, 3, 448, 448, device="cuda")
P = up.upsampler.up4.packed()
out = []
for S in (512, 256, 74):
    small = ops.adaptive_avg_pool(g[: (B if S > 100 else 2)], S, S)
    proj = ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"])
    for _ in range(2):
        kc = ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"])
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        kc = ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"])
    e.record(); torch.cuda.synchronize()
    f = "/tmp/jbuk_ref_%d.pt" % S
    if os.path.exists(f): same = torch.equal(torch.load(f), kc[:2].cpu())
    else: torch.save(kc[:2].cpu(), f); same = "ref"
    out.append(f"{S}^2: {s.elapsed_time(e)/5:.3f} ms same={same}")
print(" | ".join(out))
