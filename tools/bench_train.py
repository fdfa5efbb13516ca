#!/usr/bin/env python3
"""Forward+backward+Adam step of the probe (DINOv2-S/14, clicks injected before the backbone) at S^2.
usage: bench_train.py [B] [upsampler] [injection] [S]"""
import logging
import sys
import time
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
logging.getLogger("root").setLevel(logging.WARNING)
from helpers import S14, build_model, rand_points, seeded_
from isegprobe_amd.core.training.trainer import DataParallelTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
up = sys.argv[2] if len(sys.argv) > 2 else "bilinear"
inj = sys.argv[3] if len(sys.argv) > 3 else "before_backbone"
S = int(sys.argv[4]) if len(sys.argv) > 4 else 448
ITERS = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # simulated corrective clicks (no-grad forwards) per step
params = {"loftup": {"upsampler_path": None, "n_dim": 384}, "jbu_featup": {"backbone_type": "dinov2"},
          "lift": {"lift_path": None, "n_dim": 384, "patch": 14}}.get(up)
model = seeded_(build_model(up, injection=inj, vit=S14, img=(S, S), upsampler_params=params), 1).cuda()
torch.manual_seed(0)
image = torch.rand(B, 3, S, S, device="cuda")
gt = torch.zeros(B, 1, S, S, device="cuda")
gt[:, :, S // 4:S // 2, S // 4:3 * S // 4] = 1
points = torch.from_numpy(rand_points(np.random.default_rng(0), B, 24, S, S)).cuda()
batch = {"images": image, "instances": gt, "points": points}
trainer = DataParallelTrainer(model, lr=1e-4)
for _ in range(2):
    trainer.step(batch, num_iters=ITERS)
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    loss = trainer.step(batch, num_iters=ITERS)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"train step B={B} {up} {inj} {S}x{S} sim-clicks={ITERS}: {ms:.1f} ms/step = {B / ms * 1e3:.1f} img/s, loss {loss.item():.4f}, "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
