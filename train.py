#!/usr/bin/env python3
"""Training entry point for the probe on the HIP path (the reference's train.py:13-27 drives a
Hydra config + a model script; this one keeps the same roles with plain arguments).

    python train.py --steps 20                                   # single GPU
    python -m torch.distributed.run --nproc-per-node 8 train.py  # data parallel, RCCL over xGMI

Each rank builds the same model (frozen DINOv2 backbone + frozen upsampler, trainable embed_coords +
head), draws its own shard of the (synthetic, SBD-shaped) minibatch, and takes optimisation steps
with ONE flat-bucket gradient all-reduce per step (core/training/trainer.py).  Real datasets are
outside the dense-feature path: plug any iterable of
{"images" [B,3,H,W], "instances" [B,1,H,W], "points" [B,2P,3]} batches into DataParallelTrainer."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synthetic_batch(B, S, rng, P=24, device="cuda"):
    yy, xx = np.mgrid[:S, :S]
    images = torch.rand(B, 3, S, S)
    gts, pts = [], -np.ones((B, 2 * P, 3), np.float32)
    for b in range(B):
        cy, cx, ry, rx = rng.uniform(0.3, 0.7) * S, rng.uniform(0.3, 0.7) * S, rng.uniform(0.1, 0.3) * S, rng.uniform(0.1, 0.3) * S
        m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.float32)
        images[b] += torch.from_numpy(m)[None] * 0.5
        gts.append(torch.from_numpy(m)[None])
        pts[b, 0] = (int(cy), int(cx), 0)  # one positive click at the object centre
    return {"images": images.clamp(0, 1).to(device), "instances": torch.stack(gts).to(device),
            "points": torch.from_numpy(pts).to(device)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference train_cfg.yaml:18)")
    ap.add_argument("--size", type=int, default=224, help="crop size (reference train_cfg.yaml:22)")
    ap.add_argument("--arch", default="dinov2_vits14")
    ap.add_argument("--upsampler", default="bilinear")
    ap.add_argument("--lr", type=float, default=5e-5)
    args = ap.parse_args()

    from isegprobe_amd.core.model import iSegProbeModel
    from isegprobe_amd.core.training.trainer import DataParallelTrainer
    from isegprobe_amd.core.utils import distributed as D

    distributed = D.init_distributed()
    torch.cuda.set_device(D.get_local_rank())
    dim = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024}[args.arch]
    torch.manual_seed(0)  # identical initial weights on every rank
    model = iSegProbeModel(
        backbone_cfg={"type": "dinov2", "params": {"arch": args.arch, "feats_injection_mode": "after_backbone"}},
        head_cfg={"type": "convhead", "params": dict(in_channels=dim, num_layers=2, num_classes=1)},
        embed_coords_cfg={"type": "patchEmbed", "params": dict(img_size=(args.size, args.size), patch_size=(14, 14), embed_dim=dim)},
        upsampler_cfg={"type": args.upsampler, "params": None},
        use_disks=True, norm_radius=5, with_prev_mask=True).cuda()
    trainer = DataParallelTrainer(model, lr=args.lr)
    rng = np.random.default_rng(100 + D.get_rank())
    if D.get_rank() == 0:
        print(f"world {D.get_world_size()}  trainable bucket {trainer.bucket.nbytes() / 1e6:.1f} MB  "
              f"per-GPU batch {args.batch} @ {args.size}^2")
    for step in range(args.steps):
        batch = synthetic_batch(args.batch, args.size, rng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = trainer.step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red = D.reduce_loss_dict({"loss": loss})
        if D.get_rank() == 0:
            print(f"step {step:3d}  loss {float(red['loss']):.4f}  {dt * 1e3:7.1f} ms  "
                  f"{D.get_world_size() * args.batch / dt:7.1f} img/s")
    D.synchronize()


if __name__ == "__main__":
    main()
