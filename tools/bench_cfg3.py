#!/usr/bin/env python3
"""BASELINE configs[3]: DINOv2-L/14 + LiFT(1024) + ConvSegHead(1024), 896^2, batch 2 (flip pair), forward only."""
import logging
import sys
import time
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import build_model, rand_points, seeded_

S = int(sys.argv[1]) if len(sys.argv) > 1 else 896
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
L14 = dict(img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16)
model = seeded_(build_model("lift", vit=L14, img=(S, S), upsampler_params={"lift_path": None, "n_dim": 1024, "patch": 14}), 1).cuda().eval()
torch.manual_seed(0)
image = torch.rand(B, 4, S, S, device="cuda")
image[:, 3] = 0
points = torch.from_numpy(rand_points(np.random.default_rng(0), B, 24, S, S)).cuda()
with torch.no_grad():
    for _ in range(2):
        out = model(image, points)["instances"]
    torch.cuda.synchronize()
    n = 5
    t0 = time.perf_counter()
    for _ in range(n):
        out = model(image, points)["instances"]
    torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
assert out.shape == (B, 1, S, S) and torch.isfinite(out).all()
print(f"cfg3 L/14+LiFT {S}x{S} B={B}: {ms:.1f} ms/forward = {B / ms * 1e3:.2f} img/s; peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
