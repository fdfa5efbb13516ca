"""Shared test helpers: model construction and seeded, non-degenerate weights."""
import torch


def seeded_(module, seed, scale=1.0):
    """Same recipe as tests/golden/gen_golden.py::seeded_ (kept in sync by hand): every
    parameter and BatchNorm statistic gets a seeded, non-trivial value."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.copy_(torch.randn(p.shape, generator=g) * (scale / fan_in ** 0.5))
            elif "gamma" in name or name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, b in module.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.copy_(0.5 + torch.rand(b.shape, generator=g))
    return module


TINY_VIT = dict(img_size=70, patch_size=14, embed_dim=128, depth=2, num_heads=2)
S14 = dict(img_size=518, patch_size=14, embed_dim=384, depth=12, num_heads=6)


def build_model(upsampler="bilinear", injection="before_backbone", vit=None, img=(56, 56), upsampler_params=None,
                head_layers=2, head_type="convhead"):
    from isegprobe_amd.core.model import iSegProbeModel
    vit = vit or TINY_VIT
    D = vit["embed_dim"]
    return iSegProbeModel(
        backbone_cfg={"type": "dinov2", "params": {"arch": "custom", "feats_injection_mode": injection,
                                                   "vit_kwargs": vit}},
        head_cfg={"type": head_type, "params": (dict(in_channels=D, num_classes=1) if head_type == "linear" else
                                                dict(in_channels=D, num_layers=head_layers, num_classes=1))},
        embed_coords_cfg={"type": "patchEmbed", "params": dict(img_size=img, patch_size=(14, 14), embed_dim=D)},
        upsampler_cfg={"type": upsampler, "params": upsampler_params},
        use_disks=True, norm_radius=5, with_prev_mask=True).eval()


def rand_points(rng, B, P, H, W):
    import numpy as np
    pts = -np.ones((B, 2 * P, 3), dtype=np.float32)
    for b in range(B):
        npos, nneg = rng.integers(1, P + 1), rng.integers(0, P + 1)
        k = 0
        for pol, n in ((0, npos), (1, nneg)):
            for i in range(n):
                pts[b, pol * P + i] = (rng.integers(0, H), rng.integers(0, W), k)
                k += 1
    return pts


def seed_by_name_(module, seed, skip=()):
    """tests/golden/gen_golden.py::seed_by_name_ (kept in sync by hand): every parameter draws from its own generator
    keyed on (crc32 of its name) ^ seed, so the values do not depend on module registration order.  Used where a
    fixture cannot hold the frozen tensors (22 M DINOv2-S/14 parameters): both sides regenerate them and the fixture
    carries their sha256."""
    import zlib
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name in skip:
                continue
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) / p[0].numel() ** 0.5)
            elif "gamma" in name or name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.2 * torch.randn(p.shape, generator=g))
        for name, p in module.named_parameters():
            if name.endswith("pos_embed") and name not in skip:
                p.mul_(0.3)
    return module
