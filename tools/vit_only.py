#!/usr/bin/env python3
"""DINOv2 featurizer alone (B images at S^2, clicks fused into the patch matrix): HIP-event time per forward.
usage: vit_only.py [B] [S] [arch] [iters]   -- run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import logging
import os
import sys
import torch
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, "tests"))
logging.getLogger("root").setLevel(logging.WARNING)
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
S = int(sys.argv[2]) if len(sys.argv) > 2 else 448
arch = sys.argv[3] if len(sys.argv) > 3 else "dinov2_vits14"
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
model = bench.build("bilinear", S, arch).cuda()
image, points = bench.synthetic_batch(B, S, 1000)
image, points = image.cuda(), points.cuda()
with torch.no_grad():
    img, prev = model.prepare_input(image)
    maps = model.dist_maps(img, points)
    f = lambda: model.backbone.forward_fused_clicks(img, prev, maps, model.embed_coords)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        f()
    e.record()
    torch.cuda.synchronize()
ms = s.elapsed_time(e) / iters
vit = bench.VITS[arch]
fl = B * bench.vit_flops(vit["embed_dim"], vit["depth"], (S // 14) ** 2)
print(f"{arch} B={B} {S}x{S}: {ms:.3f} ms/forward = {fl / ms / 1e9:.0f} TFLOP/s ({fl / ms / 1e9 / 2500:.3f} of 2.5 PF)")
