#!/usr/bin/env python3
"""Where does the bf16 path's logit error come from?  (tiny golden model, GPU box)"""
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from conftest import weights_from
from helpers import build_model
from oracle import model as omodel
from oracle import vit as ovit
from isegprobe_amd.core.model._tensor import to_nchw_f32

g = dict(np.load("tests/golden/model_tiny.npz"))
w = {**weights_from(g, "common_w"), **weights_from(g, "bilinear_w")}
model = build_model("bilinear")
model.load_state_dict(w, strict=False)
model = model.cuda()
image, points = torch.from_numpy(g["image"]), torch.from_numpy(g["points"])


def stats(name, a, b):
    d = (a - b).abs()
    print(f"{name:28s} max {d.max():.4e} rms {d.pow(2).mean().sqrt():.4e}  ref rms {b.pow(2).mean().sqrt():.3f} max {b.abs().max():.3f}")


with torch.no_grad():
    img_n = omodel.normalize(image[:, :3])
    from oracle.click_maps import click_maps
    maps = torch.from_numpy(click_maps(points.numpy(), 56, 56, 5, 1.0, True))
    coord = torch.cat((image[:, 3:], maps), 1)
    clicks = ovit.patch_tokens(coord, w["embed_coords.proj.weight"], w["embed_coords.proj.bias"], 14)
    feats_ref = ovit.dinov2_features(img_n, w, patch=14, depth=2, heads=2, click_tokens=clicks,
                                     injection="before_backbone", prefix="backbone.model.")
    up_ref = F.interpolate(feats_ref, (56, 56), mode="bilinear", align_corners=True)
    logits_ref = omodel.conv_head(up_ref, w)

    img_g, prev_g = model.prepare_input(image.cuda())
    maps_g = model.dist_maps(img_g, points.cuda())
    feats_g = model.backbone.forward_fused_clicks(img_g, prev_g, maps_g, model.embed_coords)
    stats("featurizer out", to_nchw_f32(feats_g).cpu(), feats_ref)
    up_g = model.upsampler(source=feats_g, guidance=img_g)
    stats("upsampled (e2e)", to_nchw_f32(up_g).cpu(), up_ref)
    stats("logits (e2e)", model.head(up_g).cpu(), logits_ref)
    # head alone on exact (fp32 oracle) features
    stats("head on exact upsampled", model.head(up_ref.cuda()).cpu(), logits_ref)
    stats("upsample+head on exact feats", model.head(model.upsampler(source=feats_ref.cuda(), guidance=img_g)).cpu(), logits_ref)
    # pure quantisation floor: oracle head on bf16-rounded inputs/weights
    wq = {k: (v.bfloat16().float() if v.dim() == 4 and "convs" in k else v) for k, v in w.items()}
    stats("oracle head, bf16 W + bf16 in", omodel.conv_head(up_ref.bfloat16().float(), wq), logits_ref)
