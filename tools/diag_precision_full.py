#!/usr/bin/env python3
"""Where does the bf16 path's logit error come from at BASELINE sizes?  (GPU box)
    python tools/diag_precision_full.py jbu_featup 448 | loftup 224 | lift 224 | bilinear 448"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from helpers import S14, build_model, rand_points, seeded_
from oracle import model as omodel
from oracle import vit as ovit
from oracle.click_maps import click_maps
from isegprobe_amd.core.model._tensor import to_nchw_f32

up, size = sys.argv[1], int(sys.argv[2])
params = {"jbu_featup": {"backbone_type": "dinov2"}, "loftup": {"upsampler_path": None, "n_dim": 384},
          "lift": {"lift_path": None, "n_dim": 384, "patch": 14}}.get(up)
model = build_model(up, vit=S14, img=(size, size), upsampler_params=params)
seeded_(model, 321)
with torch.no_grad():
    model.backbone.model.pos_embed.mul_(0.3)
w = {k: v.clone() for k, v in model.state_dict().items()}
torch.manual_seed(11)
image = torch.rand(1, 4, size, size)
image[:, 3] = (image[:, 3] > 0.8).float()
points = torch.from_numpy(rand_points(np.random.default_rng(11), 1, 24, size, size))
cfg = dict(patch=14, depth=12, heads=6, upsampler=up, injection="before_backbone", with_prev_mask=True, use_disks=True,
           norm_radius=5)
torch.set_num_threads(16)


def stats(name, a, b):
    d = (a.float() - b.float()).abs()
    print(f"{name:44s} max {d.max():.4e} rms {d.pow(2).mean().sqrt():.4e}  rel-rms {d.pow(2).mean().sqrt() / b.pow(2).mean().sqrt():.4e}"
          f"   (ref rms {b.pow(2).mean().sqrt():.3f})", flush=True)


with torch.no_grad():
    img_n = omodel.normalize(image[:, :3])
    maps = torch.from_numpy(click_maps(points.numpy(), size, size, 5, 1.0, True))
    coord = torch.cat((image[:, 3:], maps), 1)
    clicks = ovit.patch_tokens(coord, w["embed_coords.proj.weight"], w["embed_coords.proj.bias"], 14)
    feats_ref = ovit.dinov2_features(img_n, w, patch=14, depth=12, heads=6, click_tokens=clicks,
                                     injection="before_backbone", prefix="backbone.model.")
    hr_ref, _ = omodel.features_with_grad(image, points, w, cfg)
    logits_ref = omodel.conv_head(hr_ref, w)
    # head's first layer (post-ReLU) for the split of the head's own error
    h1_ref = F.relu(F.conv2d(hr_ref, w["head.convs.0.conv.weight"], w["head.convs.0.conv.bias"], padding=1))

    model = model.cuda()
    model.fold_upsampler_affine = False
    img_g, prev_g = model.prepare_input(image.cuda())
    maps_g = model.dist_maps(img_g, points.cuda())
    feats_g = model.backbone.forward_fused_clicks(img_g, prev_g, maps_g, model.embed_coords)
    stats("featurizer out (bf16 path)", to_nchw_f32(feats_g).cpu(), feats_ref)
    stats("   floor: bf16(feats_ref)", feats_ref.bfloat16().float(), feats_ref)

    def up_and_resize(src):
        hr = model.upsampler(source=src, guidance=img_g)
        if hr.shape[2:] != img_g.shape[2:] and up != "identity":
            from isegprobe_amd import hip_ops as ops
            from isegprobe_amd.core.model._tensor import nchw_view, to_nhwc_bf16
            hr = nchw_view(ops.resize_nhwc(to_nhwc_bf16(hr), size, size, "bilinear"))
        return hr
    hr_g = up_and_resize(feats_g)
    stats("upsampled+resized (e2e)", to_nchw_f32(hr_g).cpu(), hr_ref)
    hr_x = up_and_resize(feats_ref.cuda())
    stats("upsampler on exact feats", to_nchw_f32(hr_x).cpu(), hr_ref)
    stats("   floor: bf16(hr_ref)", hr_ref.bfloat16().float(), hr_ref)
    stats("logits e2e (unfolded route)", model.head(hr_g).cpu(), logits_ref)
    stats("logits: head on exact hr", model.head(hr_ref.cuda()).cpu(), logits_ref)
    stats("logits: upsampler+head on exact feats", model.head(hr_x).cpu(), logits_ref)
    model.fold_upsampler_affine = True
    stats("logits e2e (product route)", model(image.cuda(), points.cuda())["instances"].cpu(), logits_ref)
    # quantisation floors of the head, computed by the fp32 oracle on rounded operands
    wq = {k: (v.bfloat16().float() if v.dim() == 4 and "convs" in k else v) for k, v in w.items()}
    stats("   floor: oracle head, bf16 W", omodel.conv_head(hr_ref, wq), logits_ref)
    stats("   floor: oracle head, bf16 W + bf16 in", omodel.conv_head(hr_ref.bfloat16().float(), wq), logits_ref)
    h1q = F.relu(F.conv2d(hr_ref.bfloat16().float(), wq["head.convs.0.conv.weight"], w["head.convs.0.conv.bias"], padding=1))
    h2q = F.relu(F.conv2d(h1q.bfloat16().float(), wq["head.convs.1.conv.weight"], w["head.convs.1.conv.bias"], padding=1))
    stats("   floor: + bf16 h1", F.conv2d(h2q, w["head.classifier.weight"], w["head.classifier.bias"]), logits_ref)
