#!/usr/bin/env python3
"""Per-stage bf16 error of the FeatUp-JBU product path against the device fp32 stage (csrc/jbu_f32.hip). GPU box."""
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import seeded_
from isegprobe_amd import hip_ops as ops
from isegprobe_amd.core.model.upsamplers.JBUFeatUp import JBUFeatUpUpsampler

torch.manual_seed(0)
up = seeded_(JBUFeatUpUpsampler("dinov2"), 321).cuda()
stack = up.upsampler
B, h, w, C, S = 1, 32, 32, 384, 448
x = torch.randn(B, h, w, C, device="cuda")
g = torch.randn(B, 3, S, S, device="cuda")
f32 = lambda t: t.detach().float().contiguous()


def rel(a, b):
    d = (a.float() - b.float())
    return f"max {d.abs().max():.3e} rel-rms {d.pow(2).mean().sqrt() / b.float().pow(2).mean().sqrt():.3e}"


xe = x.clone()            # exact chain (fp32)
xb = x.to(torch.bfloat16)  # product chain (bf16 in, converted exactly to f16 by the first stage)
for i, st in enumerate((stack.up1, stack.up2, stack.up3, stack.up4)):
    Bn, hh, ww, _ = xe.shape
    small = ops.adaptive_avg_pool(g, 2 * hh, 2 * ww)
    proj = ops.jbu_range_proj(small, f32(st.range_proj[0].weight.flatten(1)), f32(st.range_proj[0].bias),
                              f32(st.range_proj[3].weight.flatten(1)), f32(st.range_proj[3].bias), exact=True)
    args = (f32(st.fixup_proj[0].weight.flatten(1)), f32(st.fixup_proj[0].bias), f32(st.fixup_proj[3].weight.flatten(1)),
            f32(st.fixup_proj[3].bias), float(st.range_temp.item()), float(st.sigma_spatial.item()))
    ye = ops.jbu_stage_f32(xe, proj, small, *args)
    y_in = ops.jbu_stage_f32(xe.to(torch.float16).float(), proj, small, *args)    # input rounding only
    y1 = st.run(xe.to(torch.float16), g)                                            # product stage on the exact input (f16 inside the stack)
    yb = st.run(xb, g)                                                              # product chain
    print(f"stage {i + 1} ({2 * hh}x{2 * ww}): input-rounding only {rel(y_in, ye)} | product stage on exact input {rel(y1, ye)}"
          f" | floor f16(out) {rel(ye.to(torch.float16), ye)} | chain {rel(yb, ye)}", flush=True)
    xe, xb = ye, yb
