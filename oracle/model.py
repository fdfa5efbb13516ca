"""Oracle: the assembled iSegProbe forward (test infrastructure only).

Functional restatement of iSegBaseModel.forward (reference
core/model/iseg_base_model.py:67-89) + iSegProbeModel.backbone_forward
(core/model/iseg_probe_model.py:110-134) on a flat weight dict whose keys are the
reference model's state-dict keys (``backbone.model.*``, ``upsampler.*``,
``embed_coords.proj.*``, ``head.*``).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import upsamplers as ups
from .click_maps import click_maps
from .vit import dinov2_features, patch_tokens

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)


def normalize(image):
    """BatchImageNormalize (core/model/ops.py:96-105)."""
    m = torch.tensor(MEAN, dtype=image.dtype)[None, :, None, None]
    s = torch.tensor(STD, dtype=image.dtype)[None, :, None, None]
    return (image.clone() - m) / s


def conv_head(x, w, prefix="head.", kind="convhead"):
    """ConvSegHead / SimpleConvSegHead / SimpleClassifierHead
    (core/model/heads/conv_heads.py:10-73).  mmcv ConvModule defaults = conv+bias -> ReLU."""
    i = 0
    while f"{prefix}convs.{i}.conv.weight" in w:
        wt = w[f"{prefix}convs.{i}.conv.weight"]
        x = F.relu(F.conv2d(x, wt, w[f"{prefix}convs.{i}.conv.bias"], padding=wt.shape[-1] // 2))
        i += 1
    return F.conv2d(x, w[prefix + "classifier.weight"], w[prefix + "classifier.bias"])


def forward(image, points, w, cfg):
    """image [B,3|4,H,W] in [0,1]; points [B,2P,3].  cfg keys: patch, depth, heads,
    injection, upsampler ('identity'|'nearest'|'bilinear'|'bicubic'|'lift'|'loftup'|'jbu_featup'),
    with_prev_mask, use_disks, norm_radius; optional bn_train (batch-statistics BatchNorm in the frozen
    upsamplers, the reference's net.train()) and bn_stats_out (dict receiving the updated running statistics).
    Returns logits [B,1,H,W]."""
    with torch.no_grad():
        return forward_with_grad(image, points, w, cfg)


def forward_with_grad(image, points, w, cfg):
    """Same computation with autograd left on (weights in `w` may require grad): the oracle for the
    backward kernels."""
    hr, size = features_with_grad(image, points, w, cfg)
    logits = conv_head(hr, w)
    # iseg_base_model.py:75-80 (runs even when already H x W)
    return F.interpolate(logits, size=size, mode="bilinear", align_corners=True)


def features_with_grad(image, points, w, cfg):
    """Everything up to the head's input: (high-res feature map [B,C,H',W'] after the post-upsampler resize,
    image size).  Lets a test evaluate the head on crops when the full-size head is too slow on the CPU
    (ViT-L/14 + LiFT at 896^2: 30 TFLOP per image in the head alone)."""
    if True:
        image = image.float()
        prev = None
        if cfg.get("with_prev_mask", True):  # iseg_base_model.py:91-98
            prev, image = image[:, 3:], image[:, :3]
        image = normalize(image)
        H, W = image.shape[2:]
        maps = torch.from_numpy(click_maps(points.numpy(), H, W, cfg.get("norm_radius", 5),
                                           1.0, cfg.get("use_disks", True)))
        coord = torch.cat((prev, maps), dim=1) if prev is not None else maps  # :103-110
        clicks = patch_tokens(coord, w["embed_coords.proj.weight"], w["embed_coords.proj.bias"],
                              cfg["patch"])  # featurizers/utils/patch_embed.py:37-42
        feats = dinov2_features(image, w, patch=cfg["patch"], depth=cfg["depth"], heads=cfg["heads"],
                                click_tokens=clicks, injection=cfg.get("injection", "before_backbone"),
                                prefix="backbone.model.")
        up = cfg.get("upsampler", "bilinear")
        bn_train, stats = cfg.get("bn_train", False), cfg.get("bn_stats_out")  # net.train() semantics of the frozen upsamplers
        if up == "lift":
            hr = ups.lift(feats, image, w, "upsampler.lift.", bn_train=bn_train, stats=stats, capture=cfg.get("capture"))
        elif up == "loftup":
            hr = ups.loftup(feats, image, w, "upsampler.upsampler.", bn_train=bn_train, stats=stats)
        elif up == "jbu_featup":
            hr = ups.jbu_stack(feats, image, w, "upsampler.upsampler.", drops=cfg.get("jbu_drops"))  # net.train(): Dropout2d
        else:
            hr = getattr(ups, up)(feats, image)
        if up != "identity" and hr.shape[2:] != image.shape[2:]:  # iseg_probe_model.py:120-129
            hr = F.interpolate(hr, size=image.shape[2:], mode="bilinear", align_corners=True)
        return hr, tuple(image.shape[2:])


def nfl_loss(logits, label, alpha=0.5, gamma=2.0, eps=1e-12, ignore_label=-1):
    """NormalizedFocalLossSigmoid with the trainer's settings (core/training/losses.py:11-109, detach_delimeter=True,
    size_average=True, no max_mult; models/defaults.py builds it with alpha 0.5, gamma 2): per-sample losses [B]."""
    p = torch.sigmoid(logits)
    valid = label != ignore_label
    pt = torch.where(valid, 1.0 - (label - p).abs(), torch.ones_like(p))
    beta = (1.0 - pt) ** gamma
    mult = (valid.sum(dim=(-2, -1), keepdim=True) / (beta.sum(dim=(-2, -1), keepdim=True) + eps)).detach()
    a = torch.where(label > 0.5, alpha * valid, (1.0 - alpha) * valid)
    loss = -a * (beta * mult) * torch.log(torch.clamp(pt + eps, max=1.0)) * valid
    return loss.flatten(1).sum(1) / (valid.flatten(1).sum(1) + eps)
