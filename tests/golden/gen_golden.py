#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE's own
Python files from /root/reference (build container only; the reference never travels).

    python tests/golden/gen_golden.py            # writes tests/golden/*.npz

What is real and what is a stand-in
-----------------------------------
The reference's own files are executed unmodified: core/model/ops.py (DistMaps),
core/utils/cython/_get_dist_maps.pyx (compiled with Cython here), DINOv2.py + dinov2/layers,
featurizers/utils/patch_embed.py, upsamplers/{basic_upsamplers,LiFT,loftup/*}.py,
heads/conv_heads.py, iseg_base_model.py, iseg_probe_model.py, inference/{clicker,
transforms/*, predictors/base_predictor, evaluation}.py, inference/utils.get_iou /
compute_noc_metric.

Third-party packages that are not installed here are replaced by import stand-ins so the
files above can be imported at all.  Only three of them carry arithmetic, and those are
flagged "parity unpinned (third-party)" in oracle/__init__.py:
  * mmcv.cnn.ConvModule       -> Conv2d(bias) + ReLU (mmcv 1.6.2 defaults)
  * cv2.distanceTransform     -> scipy exact EDT
  * torchvision ToTensor      -> HWC uint8 -> CHW float/255
Everything else (wandb, omegaconf, tensorboard, timm, ftfy, easydict, albumentations,
hydra, pycocotools, ...) is inert plumbing that the path never executes.

Pretrained weights cannot be fetched (no network), so every module is instantiated with
seeded random weights, bypassing only the constructors that download
(DINOv2Featurizer.__init__ -> torch.hub, LiFT/LoftUp checkpoint loaders).
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- stand-ins
class _AnyMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _Anything(metaclass=_AnyMeta):
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return type(name, (_Anything,), {})


_STUB_ROOTS = {"wandb", "omegaconf", "timm", "ftfy", "easydict", "albumentations", "hydra",
               "pycocotools", "tensorboard", "mmcv", "cv2", "torchvision", "lvis"}


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        root = fullname.split(".")[0]
        if root in _STUB_ROOTS or fullname == "torch.utils.tensorboard":
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def install_standins():
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)

    import mmcv.cnn  # noqa: stub

    class ConvModule(nn.Module):  # mmcv 1.6.2 ConvModule defaults: conv(bias) -> ReLU, no norm
        def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
            super().__init__()
            self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
            self.activate = nn.ReLU(inplace=True)

        def forward(self, x):
            return self.activate(self.conv(x))

    mmcv.cnn.ConvModule = ConvModule

    import cv2  # noqa: stub
    from scipy.ndimage import distance_transform_edt

    cv2.DIST_L2 = 2

    def distanceTransform(mask, dist_type, mask_size):
        assert dist_type == 2 and mask_size == 0, "only exact EDT is stood in"
        return distance_transform_edt(mask).astype(np.float32)

    cv2.distanceTransform = distanceTransform

    import torchvision.transforms as T  # noqa: stub

    class ToTensor:
        def __call__(self, img):
            return torch.from_numpy(np.ascontiguousarray(img)).permute(2, 0, 1).float().div(255)

    T.ToTensor = ToTensor
    import torchvision
    torchvision.transforms = T


# --------------------------------------------------------------------------- helpers
def seeded_(module, seed, scale=1.0):
    """Seeded, non-degenerate weights: randomise every parameter AND BatchNorm statistics."""
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


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy() for k, v in module.state_dict().items()
            if "num_batches_tracked" not in k}


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB, {len(arrays)} arrays)")


def rand_points(rng, B, P, H, W, fractional=False, all_invalid_neg=False):
    pts = -np.ones((B, 2 * P, 3), dtype=np.float32)
    for b in range(B):
        npos = rng.integers(1, P + 1)
        nneg = 0 if all_invalid_neg else rng.integers(0, P + 1)
        k = 0
        for pol, n in ((0, npos), (1, nneg)):
            for i in range(n):
                r, c = rng.uniform(0, H - 1), rng.uniform(0, W - 1)
                if not fractional:
                    r, c = np.floor(r), np.floor(c)
                pts[b, pol * P + i] = (r, c, k)
                k += 1
    return pts


# --------------------------------------------------------------------------- fixtures
def gen_click_maps():
    from core.model.ops import DistMaps
    rng = np.random.default_rng(1)
    out = {}
    cases = [  # name, B, P, H, W, fractional, all_invalid_neg
        ("int_p1", 2, 1, 56, 70, False, False),
        ("int_p3", 3, 3, 56, 70, False, False),
        ("frac_p3", 3, 3, 56, 70, True, False),
        ("int_p24", 2, 24, 64, 48, False, False),
        ("noneg_p3", 2, 3, 40, 40, False, True),
        ("frac_p24_224", 1, 24, 224, 224, True, False),
    ]
    for name, B, P, H, W, frac, noneg in cases:
        pts = rand_points(rng, B, P, H, W, frac, noneg)
        out[name + "_points"] = pts
        out[name + "_hw"] = np.array([H, W])
        for disks in (True, False):
            dm = DistMaps(norm_radius=5, spatial_scale=1.0, cpu_mode=False, use_disks=disks)
            y = dm(torch.zeros(B, 3, H, W), torch.from_numpy(pts)).numpy()
            if disks:
                out[name + "_disks_bits"] = np.packbits(y.astype(np.uint8))
            else:
                out[name + "_tanh"] = y.astype(np.float32)
    save("click_maps", **out)


def gen_bfs():
    """The reference's one native component, compiled here by Cython (pyximport)."""
    from core.utils.cython import get_dist_maps
    rng = np.random.default_rng(2)
    out = {}
    cases = [("p2", 2, 30, 41), ("p5", 5, 48, 36), ("half", 3, 24, 24)]
    for name, P, H, W in cases:
        pts = rand_points(rng, 1, P, H, W)[0]
        if name == "half":  # half-integer coords pin the rounding rule (pyx:31)
            pts[0, :2] = (2.5, 3.5)
            pts[1, :2] = (7.5, 8.5)
            pts[P, :2] = (10.5, 0.5)
            pts[P, 2] = 9
        out[name + "_points"] = pts
        out[name + "_hw"] = np.array([H, W])
        for delim in (1.0, 5.0):
            out[f"{name}_d{int(delim)}"] = get_dist_maps(pts.copy(), H, W, delim)
    save("dist_maps_bfs", **out)


TINY = dict(embed_dim=128, depth=2, num_heads=2, patch=14, img_size=70)


def build_ref_backbone(injection, cfg=TINY, seed=11):
    from core.model.featurizers.DINOv2 import DinoVisionTransformer, DINOv2Featurizer
    from core.model.featurizers.dinov2.layers import MemEffAttention, NestedTensorBlock
    from functools import partial
    feat = DINOv2Featurizer.__new__(DINOv2Featurizer)  # skip torch.hub.load (DINOv2.py:491)
    nn.Module.__init__(feat)
    feat.arch = "dinov2_vits14"
    feat.feats_injection_mode = injection
    feat.model = DinoVisionTransformer(
        img_size=cfg["img_size"], patch_size=cfg["patch"], embed_dim=cfg["embed_dim"],
        depth=cfg["depth"], num_heads=cfg["num_heads"], mlp_ratio=4, init_values=1.0,
        block_chunks=0, block_fn=partial(NestedTensorBlock, attn_class=MemEffAttention))
    feat.patch_size = feat.model.patch_size
    seeded_(feat, seed)
    with torch.no_grad():
        feat.model.pos_embed.mul_(0.3)
    return feat.eval()


def gen_vit():
    rng = np.random.default_rng(3)
    torch.manual_seed(3)
    out = {}
    for injection in ("before_backbone", "after_backbone", "no_injection"):
        feat = build_ref_backbone(injection)
        for tag, (H, W) in (("sq", (56, 56)), ("rect", (42, 70)), ("native", (70, 70))):
            x = torch.randn(2, 3, H, W)
            clicks = 0.5 * torch.randn(2, (H // 14) * (W // 14), TINY["embed_dim"])
            with torch.no_grad():
                y = feat(x, clicks)
            out[f"{injection}_{tag}_x"] = x.numpy()
            out[f"{injection}_{tag}_clicks"] = clicks.numpy()
            out[f"{injection}_{tag}_y"] = y.contiguous().numpy()
        if injection == "before_backbone":
            for k, v in sd_np(feat.model).items():
                out["w::" + k] = v
            # per-stage activations of block 0 for unit tests (56x56)
            x = torch.from_numpy(out["before_backbone_sq_x"])
            with torch.no_grad():
                m = feat.model
                t = m.patch_embed(x)
                out["stage_patch_tokens"] = t.numpy()
                t = torch.cat((m.cls_token.expand(2, -1, -1), t), dim=1)
                pe = m.interpolate_pos_encoding(t, 56, 56)
                out["stage_pos_embed"] = pe.numpy()
                t = t + pe
                b0 = m.blocks[0]
                out["stage_norm1"] = b0.norm1(t).numpy()
                out["stage_attn"] = b0.attn(b0.norm1(t)).numpy()
                out["stage_block0"] = b0(t).numpy()
    save("vit_tiny", **out)


def gen_dino():
    """DINOFeaturizer ("vit" backbone type): tiny DINO-v1 ViT with seeded weights (the constructor's
    timm / torch.hub weight fetch, DINO.py:497-510, is bypassed)."""
    from functools import partial
    from core.model.featurizers.DINO import DINOFeaturizer, VisionTransformer
    torch.manual_seed(8)
    out = {}
    for feat_type in ("key", "token"):
        for inj in ("before_backbone", "after_backbone"):
            f = DINOFeaturizer.__new__(DINOFeaturizer)
            nn.Module.__init__(f)
            f.arch, f.patch_size, f.feat_type, f.feats_injection_mode, f.n_feats = "vit_small", 16, feat_type, inj, 128
            f.model = VisionTransformer(img_size=[64], patch_size=16, embed_dim=128, depth=2, num_heads=2,
                                        mlp_ratio=4, qkv_bias=True, num_classes=0,
                                        norm_layer=partial(nn.LayerNorm, eps=1e-6))
            seeded_(f, 12)
            with torch.no_grad():
                f.model.pos_embed.mul_(0.3)
            f.eval()
            x = torch.randn(2, 3, 64, 96)  # even patch-grid rows: DINO.py:205-208 drops a row+col otherwise
            clicks = 0.5 * torch.randn(2, 4 * 6, 128)
            with torch.no_grad():
                y = f(x, clicks.clone())
            tag = f"{feat_type}_{inj}"
            out[tag + "_x"], out[tag + "_clicks"], out[tag + "_y"] = x.numpy(), clicks.numpy(), y.contiguous().numpy()
    for k, v in sd_np(f.model).items():
        out["w::" + k] = v
    save("dino_tiny", **out)


def gen_simple_vit():
    from core.model.featurizers.simple_ViT import SimpleViTFeaturizer
    torch.manual_seed(9)
    f = seeded_(SimpleViTFeaturizer(image_size=(56, 84), patch_size=(14, 14), dim=128, depth=2, heads=2, mlp_dim=256,
                                    channels=3, dim_head=64), 14).eval()
    x = torch.rand(2, 3, 56, 84)
    with torch.no_grad():
        y = f(x.clone())
    out = {"x": x.numpy(), "y": y.numpy()}
    for k, v in sd_np(f).items():
        out["w::" + k] = v
    save("simple_vit_tiny", **out)


def gen_maskclip():
    """MaskCLIPFeaturizer with a tiny CLIP visual tower (clip.load's URL download, clip.py:118-177, is
    bypassed); run in fp32 on the CPU."""
    from core.model.featurizers.MaskCLIP import MaskCLIPFeaturizer
    from core.model.featurizers.maskclip.model import VisionTransformer
    torch.manual_seed(10)
    out = {}

    class _Clip(nn.Module):  # the three members MaskCLIPFeaturizer touches (maskclip/model.py:543-566)
        def __init__(self):
            super().__init__()
            self.visual = VisionTransformer(input_resolution=64, patch_size=16, width=128, layers=3, heads=2, output_dim=64)
            self.dtype = torch.float32

        def get_patch_encodings(self, image):
            return self.visual(image.type(self.dtype), patch_output=True)

        def encode_projected_patches(self, patches, orig_image_hw):
            return self.visual.forward_without_patch_embed(patches.type(self.dtype), orig_image_hw, patch_output=True)

    for inj, cdim in (("before_backbone", 128), ("after_backbone", 64), ("no_injection", 64)):
        f = MaskCLIPFeaturizer.__new__(MaskCLIPFeaturizer)
        nn.Module.__init__(f)
        f.feats_injection_mode = inj
        f.model = seeded_(_Clip(), 15).eval()
        f.patch_size = 16
        x = torch.randn(2, 3, 48, 80)
        clicks = 0.5 * torch.randn(2, 15, cdim)
        with torch.no_grad():
            y = f(x, clicks.clone())
        out[inj + "_x"], out[inj + "_clicks"], out[inj + "_y"] = x.numpy(), clicks.numpy(), y.contiguous().numpy()
    for k, v in sd_np(f.model).items():
        out["w::" + k] = v
    save("maskclip_tiny", **out)


def gen_upsamplers_and_head():
    from core.model.heads import HEAD_REGISTRY
    from core.model.upsamplers import UPSAMPLER_REGISTRY
    from core.model.upsamplers.LiFT import LiFT, LiFTUpsampler
    from core.model.upsamplers.loftup.layers import ChannelNorm
    from core.model.upsamplers.loftup.loftup import LoftUp, UpsamplerwithChannelNorm
    from core.model.upsamplers.LoftUp import LoftUpUpsampler
    torch.manual_seed(4)
    out = {}
    C, h, w, H, W = 128, 4, 5, 56, 70
    src = torch.randn(2, C, h, w)
    gd = torch.randn(2, 3, H, W)
    out["source"], out["guidance"] = src.numpy(), gd.numpy()
    src8 = src[:, :8].contiguous()  # the parameter-free upsamplers are channel-independent
    out["source8"] = src8.numpy()
    for name in ("identity", "nearest", "bilinear", "bicubic"):
        with torch.no_grad():
            out["basic_" + name] = UPSAMPLER_REGISTRY[name]()(source=src8, guidance=gd).numpy()
    # LiFT (wrapper bypasses the torch.load + .to("cuda") loader, LiFT.py:125-136)
    lift = LiFTUpsampler.__new__(LiFTUpsampler)
    nn.Module.__init__(lift)
    lift.lift = seeded_(LiFT(C, 14), 21).eval()
    with torch.no_grad():
        out["lift_y"] = lift(src, gd).numpy()
    for k, v in sd_np(lift).items():
        out["lift_w::" + k] = v
    # LoftUp (wrapper bypasses torch.load, loftup.py:152-177)
    lu = LoftUpUpsampler.__new__(LoftUpUpsampler)
    nn.Module.__init__(lu)
    lu.upsampler = UpsamplerwithChannelNorm(seeded_(LoftUp(C, lr_pe_type="sine", lr_size=16), 22),
                                            seeded_(ChannelNorm(C), 23)).eval()
    gd_small = gd[:, :, :28, :42].contiguous()
    out["loftup_guidance"] = gd_small.numpy()
    with torch.no_grad():
        out["loftup_y"] = lu(src[:, :, :2, :3].contiguous(), gd_small).numpy()
    for k, v in sd_np(lu).items():
        out["loftup_w::" + k] = v
    # heads
    x = torch.randn(2, C, 20, 24)
    out["head_x"] = x.numpy()
    for kind, kw in (("convhead", dict(in_channels=C, num_layers=2, num_classes=1)),
                     ("simple_conv", dict(in_channels=C, num_layers=2, num_classes=1)),
                     ("linear", dict(in_channels=C, num_classes=1))):
        head = seeded_(HEAD_REGISTRY[kind](**kw), 31).eval()
        with torch.no_grad():
            out[f"head_{kind}_y"] = head(x).numpy()
        for k, v in sd_np(head).items():
            out[f"head_{kind}_w::" + k] = v
    save("upsamplers_head", **out)


class _Builder:
    """Duck-typed stand-in for ModelBuilder that hands iSegProbeModel prebuilt reference
    modules (the real builder's constructors download weights)."""

    def __init__(self, backbone, upsampler, head):
        self.b, self.u, self.h = backbone, upsampler, head

    def load_featurizer(self, *a, **k):
        return self.b

    def load_upsampler(self, *a, **k):
        return self.u

    def load_head(self, *a, **k):
        return self.h


def build_ref_model(upsampler_type, injection="before_backbone", seed=40):
    from core.model.heads import ConvSegHead
    from core.model.iseg_probe_model import iSegProbeModel
    from core.model.upsamplers import UPSAMPLER_REGISTRY
    from core.model.upsamplers.LiFT import LiFT, LiFTUpsampler
    from core.model.upsamplers.loftup.layers import ChannelNorm
    from core.model.upsamplers.loftup.loftup import LoftUp, UpsamplerwithChannelNorm
    from core.model.upsamplers.LoftUp import LoftUpUpsampler
    C = TINY["embed_dim"]
    backbone = build_ref_backbone(injection, seed=seed)
    if upsampler_type == "lift":
        up = LiFTUpsampler.__new__(LiFTUpsampler)
        nn.Module.__init__(up)
        up.lift = seeded_(LiFT(C, 14), seed + 1)
    elif upsampler_type == "loftup":
        up = LoftUpUpsampler.__new__(LoftUpUpsampler)
        nn.Module.__init__(up)
        up.upsampler = UpsamplerwithChannelNorm(seeded_(LoftUp(C, lr_pe_type="sine"), seed + 2),
                                                seeded_(ChannelNorm(C), seed + 3))
    else:
        up = UPSAMPLER_REGISTRY[upsampler_type]()
    head = seeded_(ConvSegHead(C, 2, 1), seed + 4)
    model = iSegProbeModel(
        backbone_cfg={"type": "dinov2", "params": {}},
        head_cfg={"type": "convhead", "params": {}},
        embed_coords_cfg={"type": "patchEmbed", "params": {"img_size": (56, 56), "patch_size": (14, 14),
                                                             "embed_dim": C}},
        upsampler_cfg={"type": upsampler_type, "params": None},
        model_builder=_Builder(backbone, up, head),
        use_disks=True, norm_radius=5, with_prev_mask=True)
    seeded_(model.embed_coords, seed + 5)
    return model.eval()


def gen_model():
    rng = np.random.default_rng(5)
    torch.manual_seed(5)
    out = {}
    H = W = 56
    img = torch.rand(2, 4, H, W)
    img[:, 3] = (img[:, 3] > 0.7).float()
    pts = torch.from_numpy(rand_points(rng, 2, 3, H, W))
    out["image"], out["points"] = img.numpy(), pts.numpy()
    for up in ("bilinear", "identity", "lift", "loftup"):
        model = build_ref_model(up)
        with torch.no_grad():
            y = model(img, pts)["instances"]
        out[f"{up}_logits"] = y.numpy()
        # backbone / head / embed_coords are seeded identically for every variant: store once
        for k, v in sd_np(model).items():
            if k.startswith("upsampler."):
                out[f"{up}_w::" + k] = v
            else:
                key = "common_w::" + k
                assert key not in out or np.array_equal(out[key], v)
                out[key] = v
    model = build_ref_model("bilinear", injection="after_backbone")
    with torch.no_grad():
        out["bilinear_after_logits"] = model(img, pts)["instances"].numpy()
    for k, v in sd_np(model).items():
        assert np.array_equal(out["common_w::" + k], v)
    save("model_tiny", **out)


def _two_blob_scene(rng, H, W):
    """Synthetic interactive-segmentation scene: two bright ellipses on a noisy background; the TARGET is one of
    them, the other is a distractor of the same brightness -- only the clicks say which is which."""
    yy, xx = np.mgrid[:H, :W]
    while True:
        cy, cx = rng.uniform(0.25 * H, 0.75 * H, 2), rng.uniform(0.2 * W, 0.8 * W, 2)
        ry, rx = rng.uniform(0.10 * H, 0.22 * H, 2), rng.uniform(0.08 * W, 0.2 * W, 2)
        m = [((yy - cy[i]) / ry[i]) ** 2 + ((xx - cx[i]) / rx[i]) ** 2 <= 1 for i in range(2)]
        if not (m[0] & m[1]).any() and m[0].sum() > 20 and m[1].sum() > 20:
            break
    img = rng.uniform(0, 1, (H, W, 3)) * 0.25 + (m[0] | m[1])[..., None] * rng.uniform(0.45, 0.7, 3)
    return img.astype(np.float32), m[0], m[1]


def _train_click_model(model, steps=500, seed=61):
    """A few hundred CPU Adam steps of the REFERENCE tiny model (head + click patch-embed; frozen random backbone)
    on the two-blob task so that its prediction responds to clicks and straddles 0.5 (with seeded random weights
    the mask is all-positive at every click).  The trained weights are stored in the fixture."""
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    params = [p for n, p in model.named_parameters() if n.startswith(("head.", "embed_coords."))]
    for n, p in model.named_parameters():
        p.requires_grad_(n.startswith(("head.", "embed_coords.")))
    opt = torch.optim.Adam(params, lr=2e-3)
    H = W = 56
    P = 4
    for step in range(steps):
        imgs, pts, tgts = [], [], []
        for _ in range(8):
            img, tgt, other = _two_blob_scene(rng, H, W)
            pt = -np.ones((2 * P, 3), np.float32)
            # the distractor belongs to the mask until a negative click lands on it: the robot user then has
            # false positives to click on, and a negative click visibly changes the prediction
            n_neg, on_other = rng.integers(0, 3), rng.random() < 0.7
            label = tgt if (n_neg > 0 and on_other) else (tgt | other)
            k = 0
            for pol, region, n in ((0, tgt, rng.integers(1, 4)), (1, other if on_other else ~(tgt | other), n_neg)):
                rr, cc = np.nonzero(region)
                for i in range(n):
                    j = rng.integers(len(rr))
                    pt[pol * P + i] = (rr[j], cc[j], k)
                    k += 1
            mode = rng.integers(3)  # previous mask: empty / both blobs / the label
            prev = np.zeros((H, W), np.float32) if mode == 0 else ((tgt | other) if mode == 1 else label).astype(np.float32)
            imgs.append(np.concatenate([img.transpose(2, 0, 1), prev[None]], 0))
            pts.append(pt)
            tgts.append(label.astype(np.float32)[None])
        x, p, t = (torch.from_numpy(np.stack(a)) for a in (imgs, pts, tgts))
        loss = nn.functional.binary_cross_entropy_with_logits(model(x, p)["instances"], t)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 100 == 0 or step == steps - 1:
            print(f"  click-model training step {step}: bce {loss.item():.4f}")
    for p in model.parameters():
        p.requires_grad_(False)
    return model.eval()


INFERENCE_CASES = {  # tag -> ZoomIn kwargs (None: no zoom-in; the 140x196 image is a multiple of the patch size)
    "nozoom": None,
    "zoom": {"skip_clicks": -1, "target_size": (56, 56), "min_crop_size": 30},
    "zoom_skip1": {"skip_clicks": 1, "target_size": (56, 56), "min_crop_size": 30},  # reference default skip_clicks
}


def gen_inference():
    """BasePredictor fusion (flip + zoom-in + sigmoid), robot clicker, IoU, NoC -- on a model whose masks depend on
    the clicks (trained here), so that thresholds, ROI updates and the click sequence are all exercised."""
    from core.inference.clicker import Clicker
    from core.inference.evaluation import evaluate_sample
    from core.inference.predictors import get_predictor
    from core.inference.utils import compute_noc_metric, get_iou
    rng = np.random.default_rng(6)
    out = {}
    model = _train_click_model(build_ref_model("bilinear", seed=60))
    for k, v in sd_np(model).items():
        out["w::" + k] = v
    H0, W0 = 140, 196
    yy, xx = np.mgrid[:H0, :W0]
    tgt = ((yy - 62) / 30.0) ** 2 + ((xx - 120) / 38.0) ** 2 <= 1
    tgt |= ((yy - 95) / 14.0) ** 2 + ((xx - 150) / 16.0) ** 2 <= 1       # a lobe: not one clean ellipse
    other = ((yy - 80) / 24.0) ** 2 + ((xx - 42) / 26.0) ** 2 <= 1        # distractor of the same brightness
    gt = tgt.astype(np.int32)
    gt[100:110, 96:112] = -1  # ignore region
    image = (rng.uniform(0, 1, (H0, W0, 3)) * 64 + (tgt | other)[..., None] * np.array([150, 130, 120])).astype(np.uint8)
    out["image_u8"], out["gt"] = image, gt
    n_clicks = 8
    any_negative = False
    for tag, zoom in INFERENCE_CASES.items():
        predictor = get_predictor(model, "NoBRS", torch.device("cpu"), prob_thresh=0.5, zoom_in_params=zoom)
        masks, rois, near = [], [], []

        def record(image_, gt_, pred_probs, sample_id, click_indx, clicks_list):
            masks.append(np.packbits(pred_probs > 0.5))
            # pixels whose probability is within the fp32 gate of the threshold (|logit| < 1e-3 <=> |p - 0.5| < 2.5e-4):
            # the only ones an fp32-accurate implementation with another summation order may legitimately flip
            near.append(np.packbits(np.abs(pred_probs.astype(np.float64) - 0.5) < 2.5e-4))
            z = predictor.zoom_in
            rois.append((-1, -1, -1, -1) if z is None or z._object_roi is None else tuple(int(v) for v in z._object_roi))

        clicks, ious, probs = evaluate_sample(image, gt, predictor, max_iou_thr=1.01, pred_thr=0.5,
                                              max_clicks=n_clicks, callback=record)
        out[f"{tag}_clicks"] = np.array([(c.coords[0], c.coords[1], int(c.is_positive)) for c in clicks], dtype=np.int64)
        out[f"{tag}_ious"] = ious
        out[f"{tag}_probs"] = probs.astype(np.float32)
        out[f"{tag}_mask_bits"] = np.stack(masks)
        out[f"{tag}_near_bits"] = np.stack(near)
        assert all(np.unpackbits(n).sum() <= 8 for n in near), "near-threshold sets must stay tiny"
        out[f"{tag}_rois"] = np.array(rois, dtype=np.int64)
        # the fixture must not be degenerate (round-1 one had an all-positive mask and a constant IoU)
        frac = [np.unpackbits(m)[:H0 * W0].mean() for m in masks]
        print(f"  {tag}: ious {np.round(ious, 3)}  positive fraction {np.round(frac, 3)}  "
              f"negative clicks {int((out[f'{tag}_clicks'][:, 2] == 0).sum())}  distinct ROIs {len(set(rois))}")
        assert len(set(np.round(ious, 4))) >= 4, "IoU must change across clicks"
        assert all(0.01 < f < 0.95 for f in frac), "masks must straddle the threshold"
        assert ious.max() > 0.6
        any_negative = any_negative or bool((out[f"{tag}_clicks"][:, 2] == 0).any())
        if zoom is not None:
            assert len(set(r for r in rois if r[0] >= 0)) >= 2, "the zoom-in ROI must change at least once"
    assert any_negative, "at least one case must place a negative click"
    # clicker known answers
    pred = np.zeros_like(gt, dtype=bool)
    ck = Clicker(gt_mask=gt)
    seq = []
    for _ in range(4):
        ck.make_next_click(pred)
        c = ck.get_clicks()[-1]
        seq.append((c.coords[0], c.coords[1], int(c.is_positive)))
        pred = pred.copy()
        pred[max(0, c.coords[0] - 12):c.coords[0] + 12, max(0, c.coords[1] - 30):c.coords[1] + 9] = c.is_positive
    out["clicker_seq"] = np.array(seq, dtype=np.int64)
    out["clicker_final_pred"] = pred
    out["iou_value"] = np.array(get_iou(gt, pred))
    all_ious = [np.array([0.3, 0.85, 0.91, 0.95]), np.array([0.5, 0.6]), np.array([0.92])]
    noc, noc_std, over = compute_noc_metric(all_ious, [0.8, 0.85, 0.9], max_clicks=20)
    out["noc"] = np.array(noc)
    out["noc_std"] = np.array(noc_std)
    out["noc_over"] = np.array(over)
    save("inference", **out)


NOC_EVAL = dict(zoom={"skip_clicks": -1, "target_size": (56, 56)}, n_clicks=20, thresh=0.5, max_iou_thr=1.01, min_clicks=1)


def _noc_scene(rng, with_labels=False):
    """One seeded scene of the NoC fixture: a cluster of 2-5 touching ellipses of different colours on a noise
    background; the TARGET is a random subset of them, so only the clicks say which parts belong (the robot user needs
    a positive click per missed part and a negative one per wrongly included neighbour).  Returns the uint8 image and
    the object / ignore-band / distractor masks."""
    from scipy.ndimage import binary_dilation
    H, W = int(rng.integers(84, 151)), int(rng.integers(98, 201))
    yy, xx = np.mgrid[:H, :W]
    while True:
        n = int(rng.integers(2, 6))
        cen, rad = [(rng.uniform(0.4, 0.6) * H, rng.uniform(0.4, 0.6) * W)], [(rng.uniform(0.09, 0.2) * H, rng.uniform(0.08, 0.18) * W)]
        for _ in range(n - 1):  # each further ellipse leans on an earlier one
            j, a = int(rng.integers(len(cen))), rng.uniform(0, 2 * np.pi)
            r = (rng.uniform(0.07, 0.17) * H, rng.uniform(0.06, 0.15) * W)
            k = rng.uniform(0.75, 0.95)
            cen.append((cen[j][0] + k * (rad[j][0] + r[0]) * np.sin(a), cen[j][1] + k * (rad[j][1] + r[1]) * np.cos(a)))
            rad.append(r)
        lab = np.zeros((H, W), np.int64)
        for i in range(n):
            lab[((yy - cen[i][0]) / rad[i][0]) ** 2 + ((xx - cen[i][1]) / rad[i][1]) ** 2 <= 1.0] = i + 1
        member = rng.random(n) < 0.55
        member[int(rng.integers(n))] = True
        obj = np.isin(lab, 1 + np.nonzero(member)[0])
        other = (lab > 0) & ~obj
        band = binary_dilation(obj, iterations=2) & (lab == 0)
        touches = obj[0].any() or obj[-1].any() or obj[:, 0].any() or obj[:, -1].any()
        if obj.sum() > 150 and not touches and all((lab == i + 1).sum() > 40 for i in range(n)):
            break
    noise = rng.uniform(0, 64, ((H + 1) // 2, (W + 1) // 2, 3)).repeat(2, 0).repeat(2, 1)[:H, :W]  # 2x2 blocks: small PNGs
    colours = np.concatenate([np.zeros((1, 3)), rng.uniform(90, 190, (n, 3))])
    img = noise + colours[lab]
    if with_labels:  # two instances: the target group and the rest of the cluster
        return img.clip(0, 255).astype(np.uint8), obj.astype(np.uint8) + 2 * other.astype(np.uint8)
    return img.clip(0, 255).astype(np.uint8), obj, band, other


def _write_noc_tree(root, n=50, seed=12):
    """BASELINE configs[0] / SURVEY.md 8(d) "Config 0": a 50-image tree in the GrabCut on-disk layout (grabcut.py:12-42:
    data_GT/<name>.<ext>, boundary_GT/<name>.<ext>, mask values 0 / 128 = ignore band / 255 = object) of seeded scenes
    -- a target (one or two overlapping ellipses), usually a distractor of the same brightness, a noise background,
    image sizes 84..150 x 98..200.  Committed as data under tests/golden/noc_grabcut/."""
    import shutil
    from PIL import Image
    rng = np.random.default_rng(seed)
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, "data_GT")), os.makedirs(os.path.join(root, "boundary_GT"))
    for i in range(n):
        img, obj, band, other = _noc_scene(rng)
        mask = np.zeros(obj.shape, np.uint8)
        mask[obj], mask[band] = 255, 128
        ext_i, ext_m = ("png", "bmp") if i % 10 == 0 else ("png", "png")  # the real set mixes bmp / jpg / png
        Image.fromarray(img).save(os.path.join(root, "data_GT", f"{i:03d}.{ext_i}"))
        Image.fromarray(mask).save(os.path.join(root, "boundary_GT", f"{i:03d}.{ext_m}"))


def _train_noc_model(model, steps, seed=71, S=56, P=12):
    """CPU Adam training of the REFERENCE tiny model for the dataset-level fixture.  All parameters train here (in the
    reference's experiments the backbone is a pretrained DINOv2; no pretrained weights exist offline, and a random frozen
    4x4-token backbone cannot reach 90 % IoU) -- for the fixture the weights are data, stored in full.  Samples follow what
    the predictor feeds the net during evaluation: the whole image or a zoom-in crop around the object resized to SxS
    (bilinear, align_corners), click disks, previous prediction as the 4th channel; most samples take 1-6 no-grad
    interaction rounds first (next click in the largest error region), as trainer.py:399-431 does."""
    import torch.nn.functional as F
    from scipy.ndimage import distance_transform_edt, label
    rng = np.random.default_rng(seed)
    torch.manual_seed(seed)
    for p in model.parameters():
        p.requires_grad_(True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    sched = torch.optim.lr_scheduler.MultiStepLR(opt, [int(steps * 0.6), int(steps * 0.85)], 0.3)
    B = 16

    def sample():
        img, obj, band, other = _noc_scene(rng)
        H, W = obj.shape
        if rng.random() < 0.3:
            r0, r1, c0, c1 = 0, H - 1, 0, W - 1
        else:  # zoom-in crop: bounding box of (object, maybe the distractor) x 1.4 with jitter
            m = obj | (other if rng.random() < 0.4 else False)
            rr, cc = np.nonzero(m)
            h, w = (rr.max() - rr.min() + 1) * rng.uniform(1.2, 1.7), (cc.max() - cc.min() + 1) * rng.uniform(1.2, 1.7)
            cy, cx = 0.5 * (rr.min() + rr.max()) + rng.normal(0, 0.05 * h), 0.5 * (cc.min() + cc.max()) + rng.normal(0, 0.05 * w)
            r0, r1 = int(max(0, round(cy - h / 2))), int(min(H - 1, round(cy + h / 2)))
            c0, c1 = int(max(0, round(cx - w / 2))), int(min(W - 1, round(cx + w / 2)))
        t = torch.from_numpy(np.concatenate([img.astype(np.float32) / 255, obj[..., None], band[..., None], other[..., None]], 2))
        t = F.interpolate(t.permute(2, 0, 1)[None, :, r0:r1 + 1, c0:c1 + 1], size=(S, S), mode="bilinear", align_corners=True)[0]
        return t[:3], (t[3] > 0.5).numpy(), (t[4] > 0.5).numpy() & ~(t[3] > 0.5).numpy(), (t[5] > 0.5).numpy()

    def pick(region, centre):
        if centre:  # the robot user's choice: the point furthest from the region's border
            dt = distance_transform_edt(np.pad(region, 1))[1:-1, 1:-1]
            r, c = np.unravel_index(np.argmax(dt), dt.shape)
            return r, c
        rr, cc = np.nonzero(region)
        j = rng.integers(len(rr))
        return rr[j], cc[j]

    def add_click(pt, pol, rc):
        k = int((pt[:, 2] >= 0).sum())
        slot = np.nonzero(pt[pol * P:(pol + 1) * P, 2] < 0)[0]
        if len(slot):
            pt[pol * P + slot[0]] = (rc[0], rc[1], k)

    def next_click(pt, pred, obj, band):
        fn, fp = obj & ~pred, pred & ~obj & ~band
        if max(fn.sum(), fp.sum()) == 0:
            return
        pol, err = (0, fn) if fn.sum() >= fp.sum() else (1, fp)
        lab, n = label(err)
        big = lab == (1 + np.argmax([(lab == i + 1).sum() for i in range(n)]))
        add_click(pt, pol, pick(big, rng.random() < 0.7))

    for step in range(steps):
        imgs, pts, objs, bands = [], [], [], []
        for _ in range(B):
            im, obj, band, other = sample()
            if obj.sum() < 10:
                continue
            pt = -np.ones((2 * P, 3), np.float32)
            add_click(pt, 0, pick(obj, rng.random() < 0.7))
            imgs.append(im), pts.append(pt), objs.append(obj), bands.append(band)
        x = torch.cat([torch.stack(imgs), torch.zeros(len(imgs), 1, S, S)], 1)
        rounds = rng.integers(0, 7, len(imgs)) * (rng.random(len(imgs)) < 0.7)
        with torch.no_grad():
            for it in range(int(rounds.max())):
                prob = torch.sigmoid(model.eval()(x, torch.from_numpy(np.stack(pts)))["instances"])
                for b in np.nonzero(rounds > it)[0]:
                    x[b, 3] = prob[b, 0]
                    next_click(pts[b], prob[b, 0].numpy() > 0.5, objs[b], bands[b])
        model.train()
        logits = model(x, torch.from_numpy(np.stack(pts)))["instances"]
        tgt = torch.from_numpy(np.stack(objs)).float()[:, None]
        wgt = torch.from_numpy(~np.stack(bands)).float()[:, None]
        loss = (F.binary_cross_entropy_with_logits(logits, tgt, reduction="none") * wgt).sum() / wgt.sum()
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if step % 200 == 0 or step == steps - 1:
            with torch.no_grad():
                pm = logits > 0
                iou = ((pm & (tgt > 0)).flatten(1).sum(1) / ((pm | (tgt > 0)) & (wgt > 0)).flatten(1).sum(1).clamp(min=1)).mean()
            print(f"  noc-model training step {step}: bce {loss.item():.4f}  iou {iou.item():.3f}", flush=True)
    for p in model.parameters():
        p.requires_grad_(False)
    return model.eval()


def gen_noc_dataset():
    """The dataset-level half of BASELINE.json's metric ("NoC@90 parity on GrabCut"): the reference's OWN GrabCutDataset
    (core/data/datasets/grabcut.py:12-42) + evaluate_dataset (core/inference/evaluation.py:22-40) + compute_noc_metric
    (core/inference/utils.py:123-146) over a 50-image GrabCut-layout tree, driven the way evaluate.py:65-120 drives them
    (NoBRS predictor, flip on, zoom-in from the first click with a fixed target size -- eval_mode "fixed56" for the 56x56
    tiny model, get_predictor_and_zoomin_params utils.py:307-316 -- 20 clicks, thresh 0.5, print_ious -> target 1.01 so
    every click runs, utils.py:254-255).  Model: the tiny bilinear model trained by _train_noc_model (weights stored here)."""
    import cv2  # noqa: stub
    from PIL import Image
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path, flags=None: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(a[:, :, ::-1])
    root = os.path.join(OUT, "noc_grabcut")
    _write_noc_tree(root)
    steps = int(os.environ.get("NOC_TRAIN_STEPS", 2500))
    model = build_ref_model("bilinear", seed=70)
    cache = os.environ.get("NOC_MODEL_CACHE")  # developer convenience only: re-running the evaluation half without the 15-45 min of CPU training
    if cache and os.path.exists(cache):
        model.load_state_dict(torch.load(cache))
        for p_ in model.parameters():
            p_.requires_grad_(False)
        model.eval()
    else:
        model = _train_noc_model(model, steps=steps)
        if cache:
            torch.save(model.state_dict(), cache)
    out, dataset = _noc_evaluate(model, root, min_mid=26)
    # eval_cfg.yaml `clicks_limit` (inference/utils.py:286-289 -> BasePredictor.net_clicks_limit): the network sees at most 3
    # clicks of each polarity while the robot user keeps clicking -- 8 clicks per object, same tree, same model
    from core.inference.evaluation import evaluate_dataset
    from core.inference.predictors import get_predictor
    lim = get_predictor(model, "NoBRS", torch.device("cpu"), prob_thresh=NOC_EVAL["thresh"], zoom_in_params=NOC_EVAL["zoom"],
                        predictor_params={"net_clicks_limit": 3})
    lim_ious, _ = evaluate_dataset(dataset, lim, pred_thr=NOC_EVAL["thresh"], max_iou_thr=NOC_EVAL["max_iou_thr"],
                                   min_clicks=NOC_EVAL["min_clicks"], max_clicks=8)
    out["limit3_ious"] = np.stack(lim_ious).astype(np.float32)
    assert out["limit3_ious"].shape == (50, 8) and np.abs(out["limit3_ious"] - out["ious"][:, :8]).max() > 0.02  # the limit bites
    for i in range(len(dataset)):  # what the reference's reader returned: the test holds the product's reader to it
        smp = dataset.get_sample(i)
        out[f"shape_{i}"] = np.array(smp.image.shape)
        out[f"image_sum_{i}"] = np.array(smp.image.astype(np.int64).sum())
        out[f"gt_counts_{i}"] = np.array([(smp.gt_mask(smp.objects_ids[0]) == v).sum() for v in (-1, 0, 1)])
    save("noc_dataset", **out)


def _noc_evaluate(model, root, min_mid, dataset=None, mid_col=2):
    """The reference's dataset evaluation of `model` over the tree at `root` (see gen_noc_dataset) -> (arrays, dataset)."""
    from core.data.datasets.grabcut import GrabCutDataset
    from core.inference.evaluation import evaluate_dataset
    from core.inference.predictors import get_predictor
    from core.inference.utils import compute_noc_metric
    if dataset is None:
        dataset = GrabCutDataset(root)
        assert len(dataset) == 50
    n_obj = len(dataset)
    predictor = get_predictor(model, "NoBRS", torch.device("cpu"), prob_thresh=NOC_EVAL["thresh"], zoom_in_params=NOC_EVAL["zoom"])
    clicks_all, near_all = [], []

    def record(image_, gt_, pred_probs, sample_id, click_indx, clicks_list):
        if click_indx == 0:
            clicks_all.append(None), near_all.append([])
        clicks_all[-1] = [(c.coords[0], c.coords[1], int(c.is_positive)) for c in clicks_list]
        # pixels within the fp32 gate of the threshold (|logit| < 1e-3): see gen_inference
        near_all[-1].append(int((np.abs(pred_probs.astype(np.float64) - 0.5) < 2.5e-4).sum()))

    all_ious, _ = evaluate_dataset(dataset, predictor, pred_thr=NOC_EVAL["thresh"], max_iou_thr=NOC_EVAL["max_iou_thr"],
                                   min_clicks=NOC_EVAL["min_clicks"], max_clicks=NOC_EVAL["n_clicks"], callback=record)
    thrs = [0.8, 0.85, 0.9]
    noc, noc_std, over = compute_noc_metric(all_ious, thrs, max_clicks=NOC_EVAL["n_clicks"])
    ious = np.stack(all_ious)
    assert ious.shape == (n_obj, 20)
    per_obj = np.array([[(np.argmax(a >= t) + 1) if (a >= t).any() else 20 for t in thrs] for a in all_ious])
    print(f"  NoC@80/85/90 = {np.round(noc, 3)}  >=20: {over}  mIoU@1..20 = {np.round(ious.mean(0), 3)}")
    print(f"  NoC@90 per object: {per_obj[:, 2].tolist()}")
    print(f"  NoC@80 per object: {per_obj[:, 0].tolist()}")
    mid = ((per_obj[:, mid_col] > 1) & (per_obj[:, mid_col] < 20)).sum()
    assert mid >= min_mid, f"NoC@{int(thrs[mid_col] * 100)} must be neither 1 nor 20 for at least {min_mid} objects, got {mid}/{n_obj}"
    out = {"ious": ious.astype(np.float32), "noc": np.array(noc), "noc_std": np.array(noc_std), "noc_over": np.array(over),
           "noc_per_object": per_obj.astype(np.int64), "clicks": np.array(clicks_all, dtype=np.int64),
           "near_counts": np.array(near_all, dtype=np.int64), "names": np.array([str(x) for x in dataset.dataset_samples])}
    for k, v in sd_np(model).items():
        out["w::" + k] = v
    return out, dataset


def gen_noc_dataset_sbd():
    """north_star: "NoC@90 on GrabCut/SBD identical to reference".  The same kind of scenes in SBD's on-disk layout (sbd.py:79-131:
    img/<name>.jpg, inst/<name>.mat with GTinst.Segmentation, val.txt): the target group of a cluster and the rest of the cluster
    are two INSTANCES, and the reference's SBDEvaluationDataset turns each (image, instance) pair into an object -- the other
    group is the distractor; no ignore band and JPEG images, as in SBD, so 90 % IoU is rarer than on the GrabCut-layout
    tree and the non-degeneracy check is on NoC@80.  26 images -> tests/golden/noc_sbd/; the reference's evaluate_dataset + compute_noc_metric with the bilinear NoC
    model of noc_dataset.npz -> noc_dataset_sbd.npz (IoU arrays, NoC, clicks; the weights are not stored twice)."""
    import shutil
    import cv2  # noqa: stub
    from PIL import Image
    from scipy.io import savemat
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path, flags=None: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(a[:, :, ::-1])
    from core.data.datasets.sbd import SBDEvaluationDataset
    root = os.path.join(OUT, "noc_sbd")
    shutil.rmtree(root, ignore_errors=True)
    os.makedirs(os.path.join(root, "img")), os.makedirs(os.path.join(root, "inst"))
    rng = np.random.default_rng(14)
    names = [f"2008_{i:06d}" for i in range(26)]
    for name in names:
        img, lab = _noc_scene(rng, with_labels=True)
        Image.fromarray(img).save(os.path.join(root, "img", name + ".jpg"), quality=97, subsampling=0)
        savemat(os.path.join(root, "inst", name + ".mat"), {"GTinst": {"Segmentation": lab.astype(np.uint8), "Categories": np.array([[1]])}})
    open(os.path.join(root, "val.txt"), "w").write("\n".join(names) + "\n")
    base = np.load(os.path.join(OUT, "noc_dataset.npz"))
    model = build_ref_model("bilinear", seed=70)
    missing, unexpected = model.load_state_dict({k[3:]: torch.from_numpy(base[k]) for k in base.files if k.startswith("w::")}, strict=False)
    assert not unexpected and not missing, (missing, unexpected)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    model.eval()
    dataset = SBDEvaluationDataset(root, split="val")
    out, _ = _noc_evaluate(model, root, min_mid=15, dataset=dataset, mid_col=0)
    os.remove(os.path.join(root, "val_images_and_ids_list.pkl"))  # the reader's cache: every reader rebuilds it
    out = {k: v for k, v in out.items() if not k.startswith("w::")}
    out["pairs"] = np.array([[int(n.split("_")[1]), int(i)] for n, i in dataset.dataset_samples], dtype=np.int64)
    out.pop("names")
    for i in range(len(dataset)):
        smp = dataset.get_sample(i)
        out[f"image_sum_{i}"] = np.array(smp.image.astype(np.int64).sum())
        out[f"gt_count_{i}"] = np.array(int((smp.gt_mask(smp.objects_ids[0]) == 1).sum()))
    save("noc_dataset_sbd", **out)


def gen_noc_dataset_upsamplers():
    """The same dataset-level evaluation with the LEARNED upsamplers of BASELINE configs[2] / [3] (LoftUp, LiFT) in the tiny
    model: backbone / click encoder / head start from the bilinear NoC model's weights (noc_dataset.npz), the upsampler from
    its seeded initialisation, and everything is fine-tuned for NOC_FT_STEPS steps by the same recipe; then the reference's
    evaluate_dataset + compute_noc_metric run over the same 50-image tree -> noc_dataset_<upsampler>.npz."""
    import cv2  # noqa: stub
    from PIL import Image
    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path, flags=None: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(a[:, :, ::-1])
    root = os.path.join(OUT, "noc_grabcut")
    assert os.path.isdir(root), "run `gen_golden.py noc_dataset` first"
    base = np.load(os.path.join(OUT, "noc_dataset.npz"))
    steps = int(os.environ.get("NOC_FT_STEPS", 900))
    for up in os.environ.get("NOC_UPSAMPLERS", "loftup,lift").split(","):
        model = build_ref_model(up, seed=70)
        sd = {k[3:]: torch.from_numpy(base[k]) for k in base.files if k.startswith("w::")}
        missing, unexpected = model.load_state_dict(sd, strict=False)
        assert not unexpected and all(k.startswith("upsampler.") for k in missing), (missing[:5], unexpected[:5])
        cache = os.environ.get("NOC_MODEL_CACHE")
        cache = cache and f"{cache}.{up}"
        if cache and os.path.exists(cache):
            model.load_state_dict(torch.load(cache))
            for p_ in model.parameters():
                p_.requires_grad_(False)
            model.eval()
        else:
            model = _train_noc_model(model, steps=steps, seed=72)
            if cache:
                torch.save(model.state_dict(), cache)
        out, _ = _noc_evaluate(model, root, min_mid=15)
        save(f"noc_dataset_{up}", **out)


def seed_by_name_(module, seed, skip=()):
    """Seeded weights that do not depend on the order modules were registered in: every parameter draws from its own
    generator keyed on (crc32 of its name) ^ seed.  Same recipe in tests/helpers.py -- frozen tensors that are not stored
    (the 22 M DINOv2-S/14 parameters) are regenerated on the test side and verified by checksum."""
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


def gen_checkpoint():
    """SURVEY.md 8(b) "Checkpoint format": a file written by the REFERENCE's own save_checkpoint (core/utils/misc.py:36-68)
    for a model built by the reference's own constructors -- iSegProbeModel(@serialize, iseg_probe_model.py:34-46) with the
    real core.utils.model_builder.ModelBuilder (pickled into the config, serialization.py:10-38) -- the configs[0]
    architecture: DINOv2-S/14, clicks before the backbone, bilinear upsampler, ConvSegHead, 224 x 224.  Only torch.hub.load
    (DINOv2.py:491, no network) is replaced, by the same file's vit_small().  save_cfg keeps the click encoder and the
    classifier (0.9 MB; the reference's own exclude mechanism drops head.convs, which -- like the frozen backbone -- are
    regenerated from the name-keyed seed on both sides and checked by checksum).  Next to it: the reference's logits."""
    import hashlib
    from pathlib import Path
    import core.model.featurizers.DINOv2 as refdv
    from core.model.iseg_probe_model import iSegProbeModel
    from core.utils.misc import save_checkpoint
    from core.utils.model_builder import ModelBuilder
    real_hub_load = torch.hub.load
    torch.hub.load = lambda repo, arch, **kw: refdv.vit_small(patch_size=14, img_size=518, init_values=1.0, block_chunks=0)
    try:
        model = iSegProbeModel(
            backbone_cfg=dict(type="dinov2", params=dict(feats_injection_mode="before_backbone")),
            head_cfg=dict(type="convhead", params=dict(in_channels=384, num_layers=2, num_classes=1)),
            embed_coords_cfg=dict(type="patchEmbed", params=dict(img_size=(224, 224), patch_size=(14, 14), embed_dim=384)),
            neck_cfg=None,
            upsampler_cfg=dict(type="bilinear", params=None),
            save_cfg=dict(embed_coords=True, backbone=False, upsampler=False, head=dict(save=True, exclude=["convs"])),
            architecture="backbone_upsampler_head",
            model_builder=ModelBuilder(),
            use_disks=True, norm_radius=5, with_prev_mask=True)
    finally:
        torch.hub.load = real_hub_load
    seed_by_name_(model, 80)
    model.eval()
    ckpt_dir = Path(OUT) / "ref_checkpoint"
    save_checkpoint(model, ckpt_dir, verbose=False)
    path = ckpt_dir / "last_checkpoint.pth"
    saved = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(saved["state_dict"]) == ["embed_coords.proj.bias", "embed_coords.proj.weight", "head.classifier.bias", "head.classifier.weight"]
    assert type(saved["config"]["params"]["model_builder"]["value"]).__module__ == "core.utils.model_builder"
    rng = np.random.default_rng(13)
    torch.manual_seed(13)
    img = torch.rand(1, 4, 224, 224)
    img[:, 3] = (img[:, 3] > 0.8).float()
    pts = torch.from_numpy(rand_points(rng, 1, 6, 224, 224))
    torch.set_num_threads(8)
    with torch.no_grad():
        logits = model(img, pts)["instances"]
    out = {"image": img.numpy(), "points": pts.numpy(), "logits": logits.numpy(), "seed": np.array(80)}
    for k, v in saved["state_dict"].items():
        out["saved::" + k] = v.numpy()
    sha = lambda t: np.frombuffer(hashlib.sha256(t.detach().contiguous().numpy().tobytes()).digest(), np.uint8)
    for k, v in model.state_dict().items():  # every tensor the file does NOT hold: checksum of the regenerated value
        if k not in saved["state_dict"]:
            out["sha256::" + k] = sha(v)
    print(f"  {path} ({os.path.getsize(path) / 1024:.0f} KiB); logits {logits.min().item():.3f}..{logits.max().item():.3f}, "
          f"positive fraction {(logits > 0).float().mean().item():.3f}")
    save("checkpoint", **out)


def gen_frozen_checkpoints():
    """SURVEY.md 8(b) "Frozen-weight key layouts": three small files in the on-disk layouts the reference's loaders read, each
    written from the reference's own modules and READ BACK BY THE REFERENCE'S OWN LOADER before its output is stored:
      * frozen_ckpts/loftup_tiny.ckpt -- {"state_dict": {"upsampler.*", "model.1.norm.*", + keys of other sub-models that the
        loader must ignore}} -> load_loftup_checkpoint (loftup/loftup.py:152-177);
      * frozen_ckpts/lift_tiny.pth    -- flat state dict with the DataParallel "module." prefix -> load_lift_checkpoints
        (LiFT.py:125-136; its .to("cuda") is the one line skipped, there is no GPU in the build container);
      * frozen_ckpts/dinov2_tiny_hub.pth -- the DINOv2 hub layout (cls_token, pos_embed, mask_token, patch_embed.proj.*,
        blocks.{i}.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}.*, norm.*; block_chunks=0) ->
        DINOv2Featurizer with torch.hub.load replaced by the same file's class loading this state dict (DINOv2.py:491)."""
    from functools import partial
    from pathlib import Path
    import core.model.featurizers.DINOv2 as refdv
    import core.model.upsamplers.LiFT as reflift
    from core.model.featurizers.dinov2.layers import MemEffAttention, NestedTensorBlock
    from core.model.upsamplers.LoftUp import LoftUpUpsampler
    from core.model.upsamplers.loftup.layers import ChannelNorm
    from core.model.upsamplers.loftup.loftup import LoftUp
    root = Path(OUT) / "frozen_ckpts"
    root.mkdir(exist_ok=True)
    out = {}
    torch.manual_seed(17)
    C = 64
    src = torch.randn(2, C, 2, 3)
    gd = torch.randn(2, 3, 28, 42)
    out["source"], out["guidance"] = src.numpy(), gd.numpy()

    # ---- LoftUp
    up, cn = seeded_(LoftUp(C, lr_pe_type="sine", lr_size=16), 91), seeded_(ChannelNorm(C), 92)
    sd = {"upsampler." + k: v.clone() for k, v in up.state_dict().items()}
    sd.update({"model.1." + k: v.clone() for k, v in cn.state_dict().items()})
    sd["model.0.model.cls_token"] = torch.zeros(1, 1, C)  # the training wrapper's backbone lives in the same dict: ignored
    torch.save({"state_dict": sd, "epoch": 3}, root / "loftup_tiny.ckpt")
    lu = LoftUpUpsampler(str(root / "loftup_tiny.ckpt"), n_dim=C).eval()  # the reference's loader
    assert all(not p.requires_grad for p in lu.parameters())
    with torch.no_grad():
        out["loftup_y"] = lu(src, gd).numpy()

    # ---- LiFT
    lift = seeded_(reflift.LiFT(C, 14), 93)
    torch.save({"module." + k: v.clone() for k, v in lift.state_dict().items()}, root / "lift_tiny.pth")
    real_to = nn.Module.to
    nn.Module.to = lambda self, *a, **k: self if a == ("cuda",) else real_to(self, *a, **k)
    try:
        lf = reflift.LiFTUpsampler(str(root / "lift_tiny.pth"), n_dim=C, patch=14).eval()  # the reference's loader
    finally:
        nn.Module.to = real_to
    with torch.no_grad():
        out["lift_y"] = lf(src, gd).numpy()

    # ---- DINOv2, hub layout
    def tiny_vit():
        return refdv.DinoVisionTransformer(img_size=TINY["img_size"], patch_size=TINY["patch"], embed_dim=TINY["embed_dim"],
                                           depth=TINY["depth"], num_heads=TINY["num_heads"], mlp_ratio=4, init_values=1.0,
                                           block_chunks=0, block_fn=partial(NestedTensorBlock, attn_class=MemEffAttention))
    hub = seeded_(tiny_vit(), 94)
    with torch.no_grad():
        hub.pos_embed.mul_(0.3)
    torch.save({k: v.clone() for k, v in hub.state_dict().items()}, root / "dinov2_tiny_hub.pth")
    out["hub_keys"] = np.array(sorted(hub.state_dict().keys()))

    def fake_hub_load(repo, arch, **kw):
        m = tiny_vit()
        m.load_state_dict(torch.load(root / "dinov2_tiny_hub.pth", map_location="cpu"))
        return m
    real_hub_load, torch.hub.load = torch.hub.load, fake_hub_load
    try:
        feat = refdv.DINOv2Featurizer("dinov2_vits14", "before_backbone").eval()
    finally:
        torch.hub.load = real_hub_load
    img = torch.randn(2, 3, 56, 70)
    clicks = 0.3 * torch.randn(2, 4 * 5, TINY["embed_dim"])
    with torch.no_grad():
        out["dino_x"], out["dino_clicks"], out["dino_y"] = img.numpy(), clicks.numpy(), feat(img, clicks).numpy()
    for f in sorted(root.iterdir()):
        print(f"  {f} ({f.stat().st_size / 1024:.0f} KiB)")
    save("frozen_ckpts", **out)


def gen_train_step():
    """the reference iSegProbeModel in .train() (trainer.py:214 -- the frozen upsamplers'
    BatchNorm2d layers then use BATCH statistics and update their running ones), one NormalizedFocalLossSigmoid step
    (trainer.py:451-453, losses.py:11-109; alpha 0.5, gamma 2 as models/defaults.py builds it), gradients of every
    trainable tensor (embed_coords.proj.*, head.*), and the parameters after one Adam step (lr 5e-5, optimizer.py:14-35)."""
    from core.training.losses import NormalizedFocalLossSigmoid
    rng = np.random.default_rng(7)
    torch.manual_seed(7)
    out = {}
    H = W = 56
    B = 3
    img = torch.rand(B, 4, H, W)
    img[:, 3] = (img[:, 3] > 0.7).float()
    pts = torch.from_numpy(rand_points(rng, B, 3, H, W))
    gt = torch.zeros(B, 1, H, W)
    yy, xx = np.mgrid[:H, :W]
    for b in range(B):
        gt[b, 0] = torch.from_numpy((((yy - 20 - 5 * b) / 14.0) ** 2 + ((xx - 30 + 4 * b) / 18.0) ** 2 <= 1).astype(np.float32))
    gt[0, 0, 40:46, 5:15] = -1  # ignore label
    out["image"], out["points"], out["gt"] = img.numpy(), pts.numpy(), gt.numpy()
    tiny = dict(np.load(os.path.join(OUT, "model_tiny.npz")))  # the weights live there (same seed): not stored twice
    for up in ("bilinear", "lift", "loftup"):
        model = build_ref_model(up, seed=40)
        for n, p in model.named_parameters():
            p.requires_grad_(n.startswith(("head.", "embed_coords.")))
        for k, v in sd_np(model).items():
            ref = tiny[(f"{up}_w::" if k.startswith("upsampler.") else "common_w::") + k]
            assert np.array_equal(ref, v), k
        with torch.no_grad():  # the same batch in eval mode (running statistics): pinned next to the train-mode forward
            out[f"{up}_eval_logits"] = model.eval()(img, pts)["instances"].numpy()
        model.train()
        train_names = [n for n, p in model.named_parameters() if p.requires_grad]
        opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=5e-5, betas=(0.9, 0.999), eps=1e-8)
        logits = model(img, pts)["instances"]
        per_sample = NormalizedFocalLossSigmoid(alpha=0.5, gamma=2)(logits, gt)
        loss = torch.mean(per_sample)
        opt.zero_grad()
        loss.backward()
        out[f"{up}_train_logits"] = logits.detach().numpy()
        out[f"{up}_loss_per_sample"] = per_sample.detach().numpy()
        named = dict(model.named_parameters())
        for n in train_names:
            out[f"{up}_grad::" + n] = named[n].grad.numpy().copy()
        opt.step()
        for n in ("head.classifier.weight", "embed_coords.proj.bias"):  # the optimizer is torch's on both sides: two small tensors
            out[f"{up}_after_adam::" + n] = named[n].detach().numpy().copy()
        for n, b in model.named_buffers():  # running statistics after the train-mode forward (momentum 0.1)
            if n.startswith("upsampler.") and n.endswith(("running_mean", "running_var")):
                out[f"{up}_after_fwd::" + n] = b.numpy().copy()
        print(f"  {up}: loss {loss.item():.5f}  |train - eval logits| max {np.abs(out[f'{up}_train_logits'] - out[f'{up}_eval_logits']).max():.3g}")
    save("train_step", **out)


def _write_dataset_tree(root):
    """A tiny tree in every on-disk layout the reference's evaluation readers understand (lossless image files + SBD's
    .mat + PascalVOC's test pickle).  Committed under tests/golden/datasets/ as data; the arrays the reference's readers
    return for it are the fixture proper."""
    import pickle
    import shutil
    from PIL import Image
    from scipy.io import savemat
    rng = np.random.default_rng(8)
    shutil.rmtree(root, ignore_errors=True)
    img = lambda h, w: rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    # GrabCut / Berkeley layout (grabcut.py:25-38): boundary 0 / 128 (ignore) / 255 (object); mask file of another type
    g = os.path.join(root, "grabcut")
    os.makedirs(os.path.join(g, "data_GT")), os.makedirs(os.path.join(g, "boundary_GT"))
    for k, (h, w, ext_i, ext_m) in enumerate(((18, 24, "png", "bmp"), (20, 16, "bmp", "png"))):
        Image.fromarray(img(h, w)).save(os.path.join(g, "data_GT", f"s{k}.{ext_i}"))
        m = np.zeros((h, w), np.uint8)
        m[3:h - 4, 4:w - 5] = 255
        m[2, 4:w - 5] = m[h - 4, 4:w - 5] = 128
        m[5, 6] = 129 if k else 200  # any value > 128 is object (grabcut.py:38)
        Image.fromarray(np.stack([m, m, m], -1) if k else m).save(os.path.join(g, "boundary_GT", f"s{k}.{ext_m}"))
    # Berkeley (berkeley.py:6-10): the same reader on images/ + masks/
    bk = os.path.join(root, "berkeley")
    shutil.copytree(os.path.join(g, "data_GT"), os.path.join(bk, "images"))
    shutil.copytree(os.path.join(g, "boundary_GT"), os.path.join(bk, "masks"))
    # DAVIS / COCO_MVal layout (davis.py:25-38): any non-zero channel is the object
    d = os.path.join(root, "davis")
    os.makedirs(os.path.join(d, "img")), os.makedirs(os.path.join(d, "gt"))
    Image.fromarray(img(15, 21)).save(os.path.join(d, "img", "f0.png"))
    gm = np.zeros((15, 21, 3), np.uint8)
    gm[2:9, 3:12, 1], gm[10:13, 14:20, 2], gm[0, 0, 0] = 7, 200, 1
    Image.fromarray(gm).save(os.path.join(d, "gt", "f0.png"))
    # SBD (sbd.py:15-131): img/<name>.jpg, inst/<name>.mat with GTinst.Segmentation, <split>.txt
    sbd = os.path.join(root, "sbd")
    os.makedirs(os.path.join(sbd, "img")), os.makedirs(os.path.join(sbd, "inst"))
    names = ["2008_000001", "2008_000002"]
    for k, name in enumerate(names):
        flat = np.full((16, 18, 3), 30 + 60 * k, np.uint8)  # constant colour: every JPEG decoder returns the same pixels
        Image.fromarray(flat).save(os.path.join(sbd, "img", name + ".jpg"), quality=95)
        seg = np.zeros((16, 18), np.uint8)
        seg[1:6, 1:7] = 1
        seg[8:15, 4:17] = 3 + k
        if k == 0:  # a thin L-shaped "buggy" instance: area / bbox area < 0.08 (sbd.py:58-75)
            seg[7, :] = 9
            seg[15, 0] = 9
        savemat(os.path.join(sbd, "inst", name + ".mat"), {"GTinst": {"Segmentation": seg, "Categories": np.array([[1]])}})
    open(os.path.join(sbd, "val.txt"), "w").write("\n".join(names) + "\n")
    open(os.path.join(sbd, "train.txt"), "w").write(names[0] + "\n")
    # PascalVOC test split (pascalvoc.py:22-60): ImageSets/Segmentation/test.pickle = (names, instance ids)
    voc = os.path.join(root, "voc")
    os.makedirs(os.path.join(voc, "JPEGImages")), os.makedirs(os.path.join(voc, "SegmentationObject"))
    os.makedirs(os.path.join(voc, "ImageSets", "Segmentation"))
    Image.fromarray(np.full((12, 14, 3), 120, np.uint8)).save(os.path.join(voc, "JPEGImages", "a.jpg"), quality=95)
    so = np.zeros((12, 14, 3), np.uint8)
    so[1:5, 1:6] = 38       # grey 38 and 75 as instance ids, 220 the void border
    so[6:11, 5:13] = 75
    so[5, 1:6] = 220
    Image.fromarray(so).save(os.path.join(voc, "SegmentationObject", "a.png"))
    with open(os.path.join(voc, "ImageSets", "Segmentation", "test.pickle"), "wb") as f:
        pickle.dump((["a", "a"], [38, 75]), f)


def gen_datasets():
    """The reference's OWN evaluation readers (core/data/datasets/{grabcut,berkeley,davis,sbd,pascalvoc}.py + DSample) on a
    tiny tree written in their on-disk formats.  cv2 is absent: imread / cvtColor are stood in with PIL (same decoders the
    product uses; BGR2GRAY = OpenCV's documented 8-bit fixed-point weights) -- the readers' logic (ignore labels, channel
    max, GTinst indexing, instance enumeration, buggy-mask filter, pickle cache, DSample.gt_mask) is the reference's."""
    import cv2  # noqa: stub
    from PIL import Image

    def imread(path, flags=None):
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])

    cv2.COLOR_BGR2RGB, cv2.COLOR_BGR2GRAY = 4, 6

    def cvtColor(a, code):
        if code == 4:
            return np.ascontiguousarray(a[:, :, ::-1])
        b, g, r = (a[..., i].astype(np.int32) for i in range(3))
        return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.uint8)

    cv2.imread, cv2.cvtColor = imread, cvtColor
    from core.data.datasets.berkeley import BerkeleyDataset
    from core.data.datasets.davis import DavisDataset
    from core.data.datasets.grabcut import GrabCutDataset
    from core.data.datasets.pascalvoc import PascalVocDataset
    from core.data.datasets.sbd import SBDDataset, SBDEvaluationDataset
    root = os.path.join(OUT, "datasets")
    _write_dataset_tree(root)
    out = {}
    # keys = the names core/inference/utils.py:86-104 (get_dataset) accepts, built as it builds them; plus the training reader
    readers = {"GrabCut": GrabCutDataset(os.path.join(root, "grabcut")), "Berkeley": BerkeleyDataset(os.path.join(root, "berkeley")),
               "DAVIS": DavisDataset(os.path.join(root, "davis")), "COCO_MVal": DavisDataset(os.path.join(root, "davis")),
               "SBD": SBDEvaluationDataset(os.path.join(root, "sbd")),
               "SBD_Train": SBDEvaluationDataset(os.path.join(root, "sbd"), split="train"),
               "PascalVOC": PascalVocDataset(os.path.join(root, "voc"), split="test"),
               "SBDDataset_train": SBDDataset(os.path.join(root, "sbd"), split="train")}
    for name, ds in readers.items():
        out[name + "_len"] = np.array(len(ds.dataset_samples))
        for i in range(len(ds.dataset_samples)):
            smp = ds.get_sample(i)
            out[f"{name}_{i}_image"] = smp.image
            out[f"{name}_{i}_objects"] = np.array(smp.objects_ids, dtype=np.int64)
            for j in range(len(smp.objects_ids)):
                out[f"{name}_{i}_gt{j}"] = smp.gt_mask(j).astype(np.int32)
    for split in ("val", "train"):  # the caches the reader wrote: tests exercise both paths
        os.remove(os.path.join(root, "sbd", f"{split}_images_and_ids_list.pkl"))
    save("datasets", **out)


def gen_points_sampler():
    """The reference's train-time feed without augmentation: SBDDataset.__getitem__ (core/data/base_dataset.py:43-76 ->
    datasets/sbd.py:38-56 -> DSample -> MultiPointSampler.sample_object / sample_points, core/data/points_sampler.py:35-380) with
    the SBD scripts' sampler settings (models/defaults.py:74-79) on the committed SBD-layout tree, under seeded `random` /
    `numpy.random`: target masks and click lists for 40 draws per split.  cv2 is absent: imread / cvtColor as in gen_datasets,
    erode / dilate stood in with scipy's binary morphology (3 x 3 ones; erosion treats the outside as set, dilation as clear --
    OpenCV's default border values), i.e. the sampler's LOGIC and random-draw order are the reference's, the morphology
    primitives are not OpenCV's."""
    import random
    import cv2  # noqa: stub
    from PIL import Image
    from scipy import ndimage

    cv2.COLOR_BGR2RGB = 4
    cv2.imread = lambda path, flags=None: np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    cv2.cvtColor = lambda a, code: np.ascontiguousarray(a[:, :, ::-1])
    k3 = np.ones((3, 3), bool)
    cv2.erode = lambda m, kernel, iterations=1: ndimage.binary_erosion(m.astype(bool), k3, iterations=iterations, border_value=1).astype(np.uint8)
    cv2.dilate = lambda m, kernel, iterations=1: (ndimage.binary_dilation(m.astype(bool), k3, iterations=iterations, border_value=0).astype(np.uint8)
                                                  if iterations > 0 else m.copy())
    from core.data.datasets.sbd import SBDDataset
    from core.data.points_sampler import MultiPointSampler
    root = os.path.join(OUT, "datasets", "sbd")
    out = {}
    for split in ("train", "val"):
        sampler = MultiPointSampler(6, prob_gamma=0.80, merge_objects_prob=0.5, max_num_merged_objects=2)
        ds = SBDDataset(root, split=split, augmentator=None, min_object_area=20, keep_background_prob=0.01, points_sampler=sampler)
        random.seed(123), np.random.seed(123)
        pts, masks = [], []
        for _ in range(40):
            item = ds[0]
            pts.append(item["points"]), masks.append(np.packbits(item["instances"][0] > 0))
        out[f"{split}_points"], out[f"{split}_masks"] = np.stack(pts), np.stack(masks)
        out[f"{split}_shape"] = np.array(item["instances"].shape)
        out[f"{split}_image"] = (item["images"].numpy() * 255).round().astype(np.uint8)
        n_obj = len(np.unique(ds.get_sample(0)._encoded_masks)) - 1
        merged = sum(1 for m in masks if not any(np.array_equal(m, q) for q in masks[:0]))
        print(f"  {split}: {n_obj} objects, {len({m.tobytes() for m in masks})} distinct targets in 40 draws, "
              f"positives per draw {np.mean([(p[:6, 0] >= 0).sum() for p in pts]):.2f}, negatives {np.mean([(p[6:, 0] >= 0).sum() for p in pts]):.2f}")
    save("points_sampler", **out)


def gen_crops():
    """Crops transform (core/inference/transforms/crops.py): window offsets over a sweep of lengths, and one forward /
    inverse pass (crop batch, shifted clicks, overlap-averaged probabilities) per geometry."""
    from core.inference.clicker import Click
    from core.inference.transforms import Crops
    from core.inference.transforms.crops import get_offsets
    rng = np.random.default_rng(11)
    out = {}
    sweep = [(L, c, ov) for c in (32, 48, 320) for ov in (0.2, 0.35, 0.0) for L in (c, c + 1, c + 7, 2 * c - 1, 2 * c, 2 * c + 5, 3 * c + 11, 1000)]
    out["offsets_args"] = np.array(sweep, np.float64)
    flat, ptr = [], [0]
    for L, c, ov in sweep:
        flat += get_offsets(L, c, ov)
        ptr.append(len(flat))
    out["offsets_flat"], out["offsets_ptr"] = np.array(flat, np.int64), np.array(ptr, np.int64)
    cases = {"pass": ((40, 70), (48, 64), 0.2), "exact": ((48, 64), (48, 64), 0.2), "two_by_three": ((70, 150), (48, 64), 0.2),
             "tall": ((131, 64), (48, 64), 0.35)}
    for tag, ((H, W), crop, ov) in cases.items():
        image = torch.tensor(rng.standard_normal((1, 4, H, W)), dtype=torch.float32)
        clicks = [Click(True, (int(rng.integers(H)), int(rng.integers(W))), indx=0), Click(False, (3, W - 2), indx=1),
                  Click(True, (H - 1, 0), indx=2)]
        t = Crops(crop_size=crop, min_overlap=ov)
        crops, cl = t.transform(image, [clicks])
        probs = torch.tensor(rng.uniform(0, 1, (crops.shape[0], 1, *crops.shape[2:])), dtype=torch.float32)
        merged = t.inv_transform(probs)
        out[f"{tag}_args"] = np.array([H, W, crop[0], crop[1], ov], np.float64)
        out[f"{tag}_image"], out[f"{tag}_crops"] = image.numpy(), crops.numpy()
        out[f"{tag}_clicks"] = np.array([[c.is_positive, c.coords[0], c.coords[1], c.indx] for c in clicks], np.int64)
        out[f"{tag}_crop_clicks"] = np.array([[[c.is_positive, c.coords[0], c.coords[1], c.indx] for c in lst] for lst in cl], np.int64)
        out[f"{tag}_probs"], out[f"{tag}_merged"] = probs.numpy(), merged.numpy()
    save("crops", **out)


def main():
    torch.set_num_threads(4)
    install_standins()
    which = sys.argv[1:] or ["click_maps", "bfs", "vit", "dino", "simple_vit", "maskclip", "upsamplers", "model", "inference", "noc_dataset", "noc_upsamplers", "noc_sbd", "checkpoint", "frozen_ckpts", "train_step", "datasets", "points_sampler", "crops"]
    fns = {"click_maps": gen_click_maps, "bfs": gen_bfs, "vit": gen_vit, "dino": gen_dino, "simple_vit": gen_simple_vit, "maskclip": gen_maskclip,
           "upsamplers": gen_upsamplers_and_head, "model": gen_model, "inference": gen_inference, "noc_dataset": gen_noc_dataset, "noc_upsamplers": gen_noc_dataset_upsamplers, "noc_sbd": gen_noc_dataset_sbd, "checkpoint": gen_checkpoint, "frozen_ckpts": gen_frozen_checkpoints, "train_step": gen_train_step, "datasets": gen_datasets, "points_sampler": gen_points_sampler, "crops": gen_crops}
    for w in which:
        fns[w]()


if __name__ == "__main__":
    main()
