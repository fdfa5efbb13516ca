#!/usr/bin/env python3
"""Run only jbu_kernels at the 512^2 stage (for PMC passes)."""
import sys, logging
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import seeded_
from isegprobe_amd import hip_ops as ops
from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = seeded_(JBUFeatUpUpsampler("dinov2"), 3).cuda().eval()
g = torch.randn(B, 3, 448, 448, device="cuda")
P = up.upsampler.up4.packed()
small = ops.adaptive_avg_pool(g, 512, 512)
proj = ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"])
for _ in range(3):
    kc = ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"])
torch.cuda.synchronize()
