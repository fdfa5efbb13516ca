"""GPU: the assembled HIP path (featurizer / upsampler / head plugins, iSegProbeModel) against
the golden fixtures generated from the reference and against the CPU oracle.

Tolerances: the product path computes GEMMs in bf16 with fp32 accumulation, so logits are
held to BASELINE.json north_star's "within ... 1e-2 bf16", applied in the usual allclose
form |hip - ref| <= 1e-2 + 1e-2 * |ref|; masks (logits > 0) must agree wherever the oracle's
own logit is further than that tolerance from the threshold."""
import numpy as np
import pytest
import torch

from conftest import weights_from
from helpers import S14, build_model, rand_points, seeded_

pytestmark = pytest.mark.gpu
TOL = 1e-2       # BASELINE configs (full-size models): north_star's bf16 logit tolerance
TOL_TINY = 2e-2  # 2-block / 128-dim fixture models: logits are read at 4x4..56x56 pixels with no
                 # spatial averaging of the featurizer's bf16 noise (measured: max 1.4e-2, rms 3.5e-3)


def _load(model, weights):
    missing, unexpected = model.load_state_dict(weights, strict=False)
    assert not unexpected, unexpected
    assert all("mask_token" in k for k in missing), missing
    return model.cuda()


def _close(y, ref, tol=TOL):
    bad = (y - ref).abs() > tol + tol * ref.abs()
    return not bad.any().item()


def _mask_agreement(logits, ref, tol=TOL):
    decided = ref.abs() > tol
    return ((logits > 0) == (ref > 0))[decided].float().mean().item()


@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone", "no_injection"])
@pytest.mark.parametrize("tag", ["sq", "rect", "native"])
def test_featurizer_vs_golden(golden, inj, tag):
    from isegprobe_amd.core.model.featurizers import DINOv2Featurizer
    from helpers import TINY_VIT
    g = golden("vit_tiny")
    f = DINOv2Featurizer("custom", inj, vit_kwargs=TINY_VIT)
    f.model.load_state_dict(weights_from(g, "w"), strict=False)
    f = f.cuda().eval()
    y = f(torch.from_numpy(g[f"{inj}_{tag}_x"]).cuda(), torch.from_numpy(g[f"{inj}_{tag}_clicks"]).cuda())
    ref = torch.from_numpy(g[f"{inj}_{tag}_y"])
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("up", ["bilinear", "identity", "bilinear_after"])
@pytest.mark.parametrize("fused", [True, False])
def test_tiny_model_vs_golden(golden, up, fused):
    g = golden("model_tiny")
    inj = "after_backbone" if up.endswith("_after") else "before_backbone"
    model = build_model(up.replace("_after", ""), inj)
    _load(model, {**weights_from(g, "common_w"), **weights_from(g, up.replace("_after", "") + "_w")})
    image, points = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["points"]).cuda()
    with torch.no_grad():
        if fused:
            y = model(image, points)["instances"]
        else:  # plugin-by-plugin route of the reference (iseg_base_model.py:67-89)
            img, prev = model.prepare_input(image)
            coord = model.get_coord_features(img, prev, points)
            y = model.backbone_forward(img, coord)["instances"]
            y = model._to_image_size(y, img.shape[2:])
    ref = torch.from_numpy(g[up + "_logits"])
    assert y.shape == ref.shape and y.dtype == torch.float32
    err = (y.cpu() - ref).abs()
    assert _close(y.cpu(), ref, TOL_TINY), (err.max().item(), err.pow(2).mean().sqrt().item())
    assert err.pow(2).mean().sqrt().item() < 4e-3  # rms well inside the bound
    assert _mask_agreement(y.cpu(), ref, TOL_TINY) == 1.0


@pytest.mark.parametrize("size,B", [(224, 2), (448, 1)])
def test_s14_bilinear_vs_oracle(size, B):
    """Full-size DINOv2-S/14 + bilinear + ConvSegHead with seeded weights vs the CPU oracle."""
    from oracle import model as omodel
    model = build_model("bilinear", vit=S14, img=(size, size))
    seeded_(model, 123)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    w = {k: v.clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(7)
    torch.manual_seed(7)
    image = torch.rand(B, 4, size, size)
    image[:, 3] = (image[:, 3] > 0.8).float()
    points = torch.from_numpy(rand_points(rng, B, 24, size, size))
    cfg = dict(patch=14, depth=12, heads=6, upsampler="bilinear", injection="before_backbone",
               with_prev_mask=True, use_disks=True, norm_radius=5)
    torch.set_num_threads(16)
    ref = omodel.forward(image, points, w, cfg)
    with torch.no_grad():
        y = model.cuda()(image.cuda(), points.cuda())["instances"].cpu()
    err = (y - ref).abs()
    print(f"S/14@{size}: max|logit err| = {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g}, "
          f"logit range = {ref.min():.3f}..{ref.max():.3f} rms {ref.pow(2).mean().sqrt():.3f}")
    assert _close(y, ref), err.max().item()
    assert _mask_agreement(y, ref) == 1.0


@pytest.mark.parametrize("up,size,B", [("bilinear", 224, 2), ("identity", 224, 1), ("bilinear", 112, 2), ("lift", 224, 1), ("loftup", 224, 1),
                                       ("jbu_featup", 224, 1), ("jbu_featup", 448, 1)])
def test_s14_fp32_mode_vs_oracle(up, size, B):
    """north_star's fp32 gate: iSegProbeModel.forward_fp32 (fp32-accurate "three bf16 products" arithmetic on the
    same kernels) against the fp32 CPU oracle -- logits within 1e-3 (BASELINE.json configs[0]: DINOv2-S/14 + bilinear,
    fixed 224).  The bf16 product path on the same inputs is held to 1e-2 by the tests above."""
    from oracle import model as omodel
    params = {"lift": {"lift_path": None, "n_dim": 384, "patch": 14}, "loftup": {"upsampler_path": None, "n_dim": 384},
              "jbu_featup": {"backbone_type": "dinov2"}}.get(up)
    model = build_model(up, vit=S14, img=(size, size), upsampler_params=params)
    seeded_(model, 321)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    w = {k: v.clone() for k, v in model.state_dict().items()}
    rng = np.random.default_rng(11)
    torch.manual_seed(11)
    image = torch.rand(B, 4, size, size)
    image[:, 3] = (image[:, 3] > 0.8).float()
    points = torch.from_numpy(rand_points(rng, B, 24, size, size))
    cfg = dict(patch=14, depth=12, heads=6, upsampler=up, injection="before_backbone",
               with_prev_mask=True, use_disks=True, norm_radius=5)
    torch.set_num_threads(16)
    ref = omodel.forward(image, points, w, cfg)
    model = model.cuda()
    y = model.forward_fp32(image.cuda(), points.cuda())["instances"].cpu()
    assert y.shape == ref.shape and y.dtype == torch.float32
    err = (y - ref).abs()
    with torch.no_grad():
        e16 = (model(image.cuda(), points.cuda())["instances"].cpu() - ref).abs().max().item()
    print(f"fp32 mode {up}@{size}: max|logit err| = {err.max():.3g} rms {err.pow(2).mean().sqrt():.3g} "
          f"(bf16 path: {e16:.3g}); logit rms {ref.pow(2).mean().sqrt():.3f}")
    assert err.max().item() < 1e-3
    assert ((y > 0) == (ref > 0)).all() or (ref.abs()[(y > 0) != (ref > 0)] < 1e-3).all()


@pytest.mark.parametrize("up,size,params", [
    ("jbu_featup", 448, {"backbone_type": "dinov2"}),                 # BASELINE configs[1] = the bench workload
    ("loftup", 224, {"upsampler_path": None, "n_dim": 384}),           # configs[2] at the reference's crop size
    ("lift", 224, {"lift_path": None, "n_dim": 384, "patch": 14})])
def test_s14_learned_upsamplers_vs_oracle(up, size, params):
    """Full-size DINOv2-S/14 + {FeatUp JBU @448^2, LoftUp @224^2, LiFT @224^2} + ConvSegHead(384,2,1), seeded
    weights, one image, against the CPU oracle: north_star's bf16 gate |hip - ref| <= 1e-2 + 1e-2 |ref| and
    identical masks away from the threshold."""
    from oracle import model as omodel
    model = build_model(up, vit=S14, img=(size, size), upsampler_params=params)
    seeded_(model, 321)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    w = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(11)
    image = torch.rand(1, 4, size, size)
    image[:, 3] = (image[:, 3] > 0.8).float()
    points = torch.from_numpy(rand_points(np.random.default_rng(11), 1, 24, size, size))
    cfg = dict(patch=14, depth=12, heads=6, upsampler=up, injection="before_backbone",
               with_prev_mask=True, use_disks=True, norm_radius=5)
    torch.set_num_threads(16)
    ref = omodel.forward(image, points, w, cfg)
    with torch.no_grad():
        y = model.cuda()(image.cuda(), points.cuda())["instances"].cpu()
    err = (y - ref).abs()
    print(f"S/14+{up}@{size}: max|logit err| = {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g}, "
          f"logit range = {ref.min():.3f}..{ref.max():.3f} rms {ref.pow(2).mean().sqrt():.3f}, "
          f"mask agreement {_mask_agreement(y, ref):.6f}")
    assert _close(y, ref), err.max().item()
    assert _mask_agreement(y, ref) == 1.0


@pytest.mark.parametrize("fold", [True, False])
def test_tiny_jbu_model_vs_oracle(fold):
    """DINOv2(tiny) + FeatUp JBU + ConvSegHead vs the oracle, with and without folding the JBU
    fix-up affine into the head's first conv (border pixels exercise the tap table)."""
    from oracle import model as omodel
    model = build_model("jbu_featup", upsampler_params={"backbone_type": "dinov2", "feat_dim": 128})
    seeded_(model, 5)
    model.fold_upsampler_affine = fold
    w = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(1)
    image = torch.rand(2, 4, 56, 56)
    image[:, 3] = (image[:, 3] > 0.7).float()
    points = torch.from_numpy(rand_points(np.random.default_rng(3), 2, 3, 56, 56))
    cfg = dict(patch=14, depth=2, heads=2, upsampler="jbu_featup", injection="before_backbone",
               with_prev_mask=True, use_disks=True, norm_radius=5)
    ref = omodel.forward(image, points, w, cfg)
    with torch.no_grad():
        y = model.cuda()(image.cuda(), points.cuda())["instances"].cpu()
    err = (y - ref).abs()
    print(f"jbu fold={fold}: max {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g} border max "
          f"{max(err[..., 0, :].max(), err[..., -1, :].max(), err[..., :, 0].max(), err[..., :, -1].max()):.4g}")
    assert _close(y, ref, TOL_TINY), err.max().item()
    assert _mask_agreement(y, ref, TOL_TINY) == 1.0


def test_tiny_loftup_model_vs_golden(golden):
    g = golden("model_tiny")
    model = build_model("loftup", upsampler_params={"upsampler_path": None, "n_dim": 128})
    missing, unexpected = model.load_state_dict({**weights_from(g, "common_w"), **weights_from(g, "loftup_w")}, strict=False)
    assert not unexpected and all(("mask_token" in k or "num_batches_tracked" in k) for k in missing), missing
    with torch.no_grad():
        y = model.cuda()(torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["points"]).cuda())["instances"].cpu()
    ref = torch.from_numpy(g["loftup_logits"])
    err = (y - ref).abs()
    print(f"loftup model: max {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g} ref rms {ref.pow(2).mean().sqrt():.3f}")
    assert _close(y, ref, TOL_TINY), err.max().item()
    assert _mask_agreement(y, ref, TOL_TINY) == 1.0


def test_tiny_lift_model_vs_golden(golden):
    g = golden("model_tiny")
    model = build_model("lift", upsampler_params={"lift_path": None, "n_dim": 128, "patch": 14})
    missing, unexpected = model.load_state_dict({**weights_from(g, "common_w"), **weights_from(g, "lift_w")}, strict=False)
    assert not unexpected and all(("mask_token" in k or "num_batches_tracked" in k) for k in missing), missing
    with torch.no_grad():
        y = model.cuda()(torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["points"]).cuda())["instances"].cpu()
    ref = torch.from_numpy(g["lift_logits"])
    err = (y - ref).abs()
    print(f"lift model: max {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g} ref rms {ref.pow(2).mean().sqrt():.3f}")
    assert _close(y, ref, TOL_TINY), err.max().item()
    assert _mask_agreement(y, ref, TOL_TINY) == 1.0


@pytest.mark.parametrize("up", ["bilinear", "identity", "bilinear_after", "lift", "loftup"])
def test_fp32_mode_vs_reference_golden(golden, up):
    """forward_fp32 against logits produced by the REFERENCE's own code (tests/golden/gen_golden.py, fp32 on CPU):
    north_star's "logits within 1e-3 fp32" on the fixture models, for every upsampler the fp32 mode covers and both
    click-injection points."""
    g = golden("model_tiny")
    base = up.replace("_after", "")
    params = {"lift": {"lift_path": None, "n_dim": 128, "patch": 14}, "loftup": {"upsampler_path": None, "n_dim": 128}}.get(base)
    model = build_model(base, "after_backbone" if up.endswith("_after") else "before_backbone", upsampler_params=params)
    missing, unexpected = model.load_state_dict({**weights_from(g, "common_w"), **weights_from(g, base + "_w")}, strict=False)
    assert not unexpected and all(("mask_token" in k or "num_batches_tracked" in k) for k in missing), missing
    y = model.cuda().forward_fp32(torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["points"]).cuda())["instances"].cpu()
    ref = torch.from_numpy(g[up + "_logits"])
    assert y.shape == ref.shape
    err = (y - ref).abs()
    print(f"fp32 mode vs reference golden [{up}]: max {err.max():.3g} rms {err.pow(2).mean().sqrt():.3g} "
          f"(ref rms {ref.pow(2).mean().sqrt():.3f})")
    assert err.max().item() < 1e-3


@pytest.mark.parametrize("feat_type", ["key", "token"])
@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone"])
def test_dino_vit_featurizer_vs_golden(golden, feat_type, inj):
    from isegprobe_amd.core.utils.model_builder import ModelBuilder
    g = golden("dino_tiny")
    f = ModelBuilder().load_featurizer("vit", dict(arch="vit_small", patch_size=16, feat_type=feat_type, feats_injection_mode=inj,
                                                   vit_kwargs=dict(img_size=64, embed_dim=128, depth=2, num_heads=2)))
    f.model.load_state_dict(weights_from(g, "w"))
    f = f.cuda().eval()
    tag = f"{feat_type}_{inj}"
    with torch.no_grad():
        y = f(torch.from_numpy(g[tag + "_x"]).cuda(), torch.from_numpy(g[tag + "_clicks"]).cuda())
    ref = torch.from_numpy(g[tag + "_y"])
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone", "no_injection"])
@pytest.mark.parametrize("tag", ["sq", "rect", "native"])
def test_featurizer_fp32_mode_vs_reference_golden(golden, inj, tag):
    """DINOv2 featurizer in the fp32 checking mode against the reference-generated features (1e-3 gate)."""
    from isegprobe_amd.core.model.featurizers import DINOv2Featurizer
    from isegprobe_amd.core.model.precise import featurizer_fp32
    from helpers import TINY_VIT
    g = golden("vit_tiny")
    f = DINOv2Featurizer("custom", inj, vit_kwargs=TINY_VIT)
    f.model.load_state_dict(weights_from(g, "w"), strict=False)
    f = f.cuda().eval()
    y = featurizer_fp32(f, torch.from_numpy(g[f"{inj}_{tag}_x"]).cuda(), torch.from_numpy(g[f"{inj}_{tag}_clicks"]).cuda()).cpu()
    ref = torch.from_numpy(g[f"{inj}_{tag}_y"])
    assert y.shape == ref.shape
    assert (y - ref).abs().max().item() < 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("feat_type", ["key", "token"])
@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone"])
def test_dino_vit_featurizer_fp32_mode_vs_reference_golden(golden, feat_type, inj):
    """DINO ViT-S/16 featurizer (last-block keys or tokens) in the fp32 checking mode vs the reference-generated features."""
    from isegprobe_amd.core.model.precise import featurizer_fp32
    from isegprobe_amd.core.utils.model_builder import ModelBuilder
    g = golden("dino_tiny")
    f = ModelBuilder().load_featurizer("vit", dict(arch="vit_small", patch_size=16, feat_type=feat_type, feats_injection_mode=inj,
                                                   vit_kwargs=dict(img_size=64, embed_dim=128, depth=2, num_heads=2)))
    f.model.load_state_dict(weights_from(g, "w"))
    f = f.cuda().eval()
    tag = f"{feat_type}_{inj}"
    y = featurizer_fp32(f, torch.from_numpy(g[tag + "_x"]).cuda(), torch.from_numpy(g[tag + "_clicks"]).cuda()).cpu()
    ref = torch.from_numpy(g[tag + "_y"])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    print(f"DINO fp32 mode [{tag}]: max err {err:.3g} (ref max {ref.abs().max():.3g})")
    assert err < 1e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone", "no_injection"])
def test_maskclip_featurizer_fp32_mode_vs_reference_golden(golden, inj):
    from isegprobe_amd.core.model.precise import maskclip_featurizer_fp32
    from isegprobe_amd.core.utils.model_builder import ModelBuilder
    g = golden("maskclip_tiny")
    if inj + "_x" not in g:
        pytest.skip("no fixture for this injection mode")
    f = ModelBuilder().load_featurizer("mask_clip", dict(model_name="tiny", feats_injection_mode=inj,
                                                          visual_kwargs=dict(input_resolution=64, patch_size=16, width=128,
                                                                             layers=3, heads=2, output_dim=64)))
    f.model.load_state_dict(weights_from(g, "w"))
    f = f.cuda().eval()
    y = maskclip_featurizer_fp32(f, torch.from_numpy(g[inj + "_x"]).cuda(), torch.from_numpy(g[inj + "_clicks"]).cuda()).cpu()
    ref = torch.from_numpy(g[inj + "_y"])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    print(f"MaskCLIP fp32 mode [{inj}]: max err {err:.3g} (ref max {ref.abs().max():.3g})")
    assert err < 1e-3 * max(1.0, ref.abs().max().item())


def test_simple_vit_click_encoder_vs_golden(golden):
    from isegprobe_amd.core.utils.model_builder import ModelBuilder
    g = golden("simple_vit_tiny")
    f = ModelBuilder().load_featurizer("simple_vit", dict(img_size=(56, 84), patch_size=(14, 14), embed_dim=128, depth=2,
                                                          heads=2, mlp_dim=256, channels=3, dim_head=64), freeze=True)
    f.load_state_dict(weights_from(g, "w"))
    with torch.no_grad():
        y = f.cuda().eval()(torch.from_numpy(g["x"]).cuda()).cpu()
    ref = torch.from_numpy(g["y"])
    assert y.shape == ref.shape
    err = (y - ref).abs().max().item()
    print("simple_vit max err", err, "ref max", ref.abs().max().item())
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("inj", ["before_backbone", "after_backbone", "no_injection"])
def test_maskclip_featurizer_vs_golden(golden, inj):
    from isegprobe_amd.core.utils.model_builder import ModelBuilder
    g = golden("maskclip_tiny")
    f = ModelBuilder().load_featurizer("mask_clip", dict(model_name="tiny", feats_injection_mode=inj,
                                                          visual_kwargs=dict(input_resolution=64, patch_size=16, width=128,
                                                                             layers=3, heads=2, output_dim=64)))
    f.model.load_state_dict(weights_from(g, "w"))
    with torch.no_grad():
        y = f.cuda().eval()(torch.from_numpy(g[inj + "_x"]).cuda(), torch.from_numpy(g[inj + "_clicks"]).cuda())
    ref = torch.from_numpy(g[inj + "_y"])
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err < 3e-2 * max(1.0, ref.abs().max().item()), err
