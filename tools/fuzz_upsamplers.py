#!/usr/bin/env python3
"""Random-size sweep of the upsampler plugins against the CPU oracle (GPU box): FeatUp JBU (plain stack, and the stages fused
with the resize where 16h x 16w -> 14h x 14w applies), LoftUp (half-precision inference stream), LiFT.
usage: fuzz_upsamplers.py [seed] [rounds]"""
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import seeded_
from isegprobe_amd.core.model._tensor import to_nchw_f32
from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler, LiFTUpsampler, LoftUpUpsampler
from oracle import upsamplers as oups

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
g = torch.Generator().manual_seed(seed)
ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
bad = 0
torch.set_num_threads(16)


def check(name, got, ref, tol_max, tol_rms):
    global bad
    err = (got - ref).abs()
    scale, rms = max(1.0, ref.abs().max().item()), max(1.0, ref.pow(2).mean().sqrt().item())
    ok = err.max().item() <= tol_max * scale and err.pow(2).mean().sqrt().item() <= tol_rms * rms
    print(f"{'ok  ' if ok else 'FAIL'} {name}: max {err.max().item():.3g} rms {err.pow(2).mean().sqrt().item():.3g} (ref max {ref.abs().max().item():.3g})", flush=True)
    bad += 0 if ok else 1


for r in range(rounds):
    torch.manual_seed(seed * 100 + r)
    B, h, w = ri(1, 2), ri(2, 8), ri(2, 8)
    C = [64, 128][ri(0, 1)]
    src = torch.randn(B, C, h, w)
    # ---- FeatUp JBU: arbitrary guidance size (plain stack), and the patch-14 image size (last stage fused with the resize)
    up = seeded_(JBUFeatUpUpsampler("dinov2", feat_dim=C), seed + r)
    wsd = {k: v.clone() for k, v in up.state_dict().items()}
    up = up.cuda()
    gd = torch.randn(B, 3, ri(16 * h, 20 * h), ri(16 * w, 20 * w))
    with torch.no_grad():
        y = to_nchw_f32(up(src.cuda(), gd.cuda())).cpu()
    check(f"jbu stack B{B} {h}x{w} C{C} guidance {tuple(gd.shape[2:])}", y, oups.jbu_stack(src, gd, wsd, "upsampler."), 3e-2, 5e-3)
    gd14 = torch.randn(B, 3, 14 * h, 14 * w)
    with torch.no_grad():
        fused = to_nchw_f32(up.upsampler.forward_stages(src.cuda(), gd14.cuda(), out_size=(14 * h, 14 * w))).cuda()
        conv = up.upsampler.fixup_proj[1]
        full = (fused + 0.1 * F.conv2d(fused, conv.weight.float(), conv.bias.float())).cpu()
    ref = F.interpolate(oups.jbu_stack(src, gd14, wsd, "upsampler."), (14 * h, 14 * w), mode="bilinear", align_corners=True)
    check(f"jbu fused resize B{B} {h}x{w} C{C}", full, ref, 3e-2, 5e-3)
    # ---- LoftUp (n_dim 128 -> head_dim 37 -> 64; n_dim 384 -> 101 -> 128)
    nd = [128, 384][ri(0, 1)]
    lu = seeded_(LoftUpUpsampler(upsampler_path=None, n_dim=nd), seed + r)
    wl = {k: v.clone() for k, v in lu.state_dict().items()}
    lu = lu.cuda().eval()
    H, W = 14 * ri(2, 5), 14 * ri(2, 5)
    s2, g2 = torch.randn(B, nd, H // 14, W // 14), torch.rand(B, 3, H, W)
    with torch.no_grad():
        y = to_nchw_f32(lu(s2.cuda(), g2.cuda())).cpu()
    check(f"loftup n_dim {nd} B{B} {H}x{W}", y, oups.loftup(s2, g2, wl, "upsampler."), 2e-2, 4e-3)
    # ---- LiFT
    lf = seeded_(LiFTUpsampler(lift_path=None, n_dim=nd, patch=14), seed + r)
    wf = {k: v.clone() for k, v in lf.state_dict().items()}
    lf = lf.cuda().eval()
    with torch.no_grad():
        y = to_nchw_f32(lf(s2.cuda(), g2.cuda())).cpu()
    check(f"lift n_dim {nd} B{B} {H}x{W}", y, oups.lift(s2, g2, wf, "lift."), 3e-2, 6e-3)
print(f"{bad} mismatches")
sys.exit(1 if bad else 0)
