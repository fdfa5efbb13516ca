"""GPU: upsampler plugins (HIP) against the oracle / golden fixtures."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import weights_from
from helpers import seeded_

pytestmark = pytest.mark.gpu


def _f32(x):
    from isegprobe_amd.core.model._tensor import to_nchw_f32
    return to_nchw_f32(x).cpu()


@pytest.mark.parametrize("name", ["identity", "nearest", "bilinear", "bicubic"])
def test_basic_vs_golden(golden, name):
    from isegprobe_amd.core.model.upsamplers import UPSAMPLER_REGISTRY
    g = golden("upsamplers_head")
    src = torch.from_numpy(g["source8"]).cuda()
    y = UPSAMPLER_REGISTRY[name]()(source=src, guidance=torch.from_numpy(g["guidance"]).cuda())
    ref = torch.from_numpy(g["basic_" + name])
    assert tuple(y.shape) == tuple(ref.shape)
    # inputs are rounded to bf16 on entry and outputs stored as bf16: 2^-8 relative
    assert (_f32(y) - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("size", [(448, 448, 64, 64), (448, 448, 128, 128), (448, 448, 512, 512), (56, 70, 8, 10)])
def test_adaptive_avg_pool(size):
    from isegprobe_amd import hip_ops as ops
    H, W, OH, OW = size
    x = torch.randn(2, 3, H, W, device="cuda")
    assert (ops.adaptive_avg_pool(x, OH, OW) - F.adaptive_avg_pool2d(x, (OH, OW))).abs().max().item() < 1e-5


@pytest.mark.parametrize("heads", ["convhead", "simple_conv", "linear"])
def test_heads_vs_golden(golden, heads):
    from isegprobe_amd.core.model.heads import HEAD_REGISTRY
    g = golden("upsamplers_head")
    kw = dict(in_channels=128, num_classes=1) if heads == "linear" else dict(in_channels=128, num_layers=2, num_classes=1)
    head = HEAD_REGISTRY[heads](**kw)
    head.load_state_dict(weights_from(g, f"head_{heads}_w"))
    y = head.cuda()(torch.from_numpy(g["head_x"]).cuda()).cpu()
    ref = torch.from_numpy(g[f"head_{heads}_y"])
    assert y.shape == ref.shape
    assert ((y - ref).abs() <= 1e-2 + 1e-2 * ref.abs()).all()


def test_jbu_stages_and_stack_vs_oracle():
    from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
    from oracle import upsamplers as oups
    torch.manual_seed(0)
    C = 128
    up = JBUFeatUpUpsampler("dinov2", feat_dim=C)
    seeded_(up, 77)
    with torch.no_grad():
        for s in range(1, 5):
            st = getattr(up.upsampler, f"up{s}")
            st.range_temp.fill_(0.3 * s)
            st.sigma_spatial.fill_(0.8 + 0.1 * s)
    w = {k: v.clone() for k, v in up.state_dict().items()}
    src = torch.randn(2, C, 4, 5)
    gd = torch.randn(2, 3, 64, 80)
    ref = oups.jbu_stack(src, gd, w, "upsampler.")
    y = _f32(up.cuda()(src.cuda(), gd.cuda()))
    assert tuple(y.shape) == (2, C, 64, 80)
    err = (y - ref).abs()
    print("jbu max err", err.max().item(), "rms", err.pow(2).mean().sqrt().item(), "ref rms", ref.pow(2).mean().sqrt().item())
    assert err.max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 5e-3 * max(1.0, ref.pow(2).mean().sqrt().item())


def test_jbu_train_mode_dropout():
    """The reference trains with net.train() on the whole model (trainer.py:214), which switches on the frozen FeatUp
    stack's Dropout2d layers.  With pinned multipliers the stack equals the oracle's stack with the same multipliers; all-
    ones multipliers reproduce the eval output bit for bit; an eval forward never drops anything; and two train-mode
    forwards differ (fresh draws)."""
    from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
    from oracle import upsamplers as oups
    torch.manual_seed(1)
    C = 128
    up = seeded_(JBUFeatUpUpsampler("dinov2", feat_dim=C), 21)
    w = {k: v.clone() for k, v in up.state_dict().items()}
    src, gd = torch.randn(2, C, 4, 5), torch.randn(2, 3, 64, 80)
    up = up.cuda()
    stack = up.upsampler
    drops = stack.draw_dropout(2, "cuda", torch.Generator(device="cuda").manual_seed(3))
    assert all((d == 0).any() for st in drops["stages"] for d in (st[0], st[1][:, :49])) and (drops["fixup"] == 0).any()
    cpu = {"stages": [(r.cpu(), f[:, :49].cpu()) for r, f in drops["stages"]], "fixup": drops["fixup"].cpu()}
    ref = oups.jbu_stack(src, gd, w, "upsampler.", drops=cpu)
    ref_eval = oups.jbu_stack(src, gd, w, "upsampler.")
    assert (ref - ref_eval).abs().max().item() > 0.05  # the draw matters
    with torch.no_grad():
        y = _f32(up(src.cuda(), gd.cuda(), drops=drops))
        y_eval = _f32(up(src.cuda(), gd.cuda()))
        ones = {"stages": [(torch.ones(2, 32, device="cuda"), torch.ones(2, 64, device="cuda"))] * 4,
                "fixup": torch.ones(2, C, device="cuda")}
        assert torch.equal(_f32(up(src.cuda(), gd.cuda(), drops=ones)), y_eval)
    for got, want in ((y, ref), (y_eval, ref_eval)):
        err = (got - want).abs()
        assert err.max().item() < 3e-2 * max(1.0, want.abs().max().item())
        assert err.pow(2).mean().sqrt().item() < 5e-3 * max(1.0, want.pow(2).mean().sqrt().item())
    # train mode draws fresh multipliers per forward, with or without autograd (nn.Dropout2d's rule); eval mode never does
    up.train()
    xs = src.cuda().requires_grad_(True)
    a, b = _f32(up(xs, gd.cuda())), _f32(up(xs, gd.cuda()))
    assert (a - b).abs().max().item() > 1e-3
    with torch.no_grad():
        assert (_f32(up(src.cuda(), gd.cuda())) - y_eval).abs().max().item() > 1e-3
    up.eval()
    assert torch.equal(_f32(up(xs, gd.cuda())).detach(), y_eval)
    # gradient through the dropped fix-up: d/dx sum(c * stack(x)) against the oracle's autograd with the same draw
    up.train()
    stack.fixed_dropout = drops
    coef = torch.randn(2, C, 64, 80)
    xb = src.cuda().permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).requires_grad_(True)  # the featurizer's output form
    (up(xb.permute(0, 3, 1, 2), gd.cuda()).float() * coef.cuda()).sum().backward()
    xr = xb.detach().float().permute(0, 3, 1, 2).cpu().requires_grad_(True)
    (oups.jbu_stack(xr, gd, w, "upsampler.", drops=cpu) * coef).sum().backward()
    g, gr = xb.grad.float().permute(0, 3, 1, 2).cpu(), xr.grad
    assert torch.nn.functional.cosine_similarity(g.flatten(), gr.flatten(), dim=0).item() > 0.999
    assert (g - gr).pow(2).mean().sqrt().item() < 2e-2 * gr.pow(2).mean().sqrt().item()


@pytest.mark.parametrize("h,w", [(4, 8), (8, 4), (16, 16), (5, 7), (3, 9)])
def test_jbu_last_stage_fused_with_resize(h, w):
    """JBUStack.forward_stages(out_size=image size): the last x2 stage and the model's bilinear resize as ONE operator
    (isp_jbu_blend + isp_jbu_apply_resized) against stage -> isp_resize (both bf16 paths) and against the fp32 oracle
    stack followed by F.interpolate; sizes whose ratio is not 8:7 fall back to the separate resize."""
    import torch.nn.functional as F
    from isegprobe_amd import hip_ops as ops
    from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
    from isegprobe_amd.core.model.upsamplers.JBUFeatUp import JBULearnedRange
    from isegprobe_amd.core.model._tensor import nchw_view, to_nhwc_bf16
    from oracle import upsamplers as oups
    torch.manual_seed(h * 31 + w)
    C = 64
    up = JBUFeatUpUpsampler("dinov2", feat_dim=C)
    seeded_(up, 5)
    w_sd = {k: v.clone() for k, v in up.state_dict().items()}
    H, W = 14 * h, 14 * w                      # patch-14 image of an h x w token grid; FeatUp's map is 16h x 16w
    src, gd = torch.randn(2, C, h, w), torch.randn(2, 3, H, W)
    stack = up.cuda().upsampler
    assert JBULearnedRange.resize_fusable(16 * h, 16 * w, H, W) and not JBULearnedRange.resize_fusable(16 * h, 16 * w, H - 1, W)
    fused = _f32(stack.forward_stages(src.cuda(), gd.cuda(), out_size=(H, W)))
    assert tuple(fused.shape) == (2, C, H, W)
    plain = stack.forward_stages(src.cuda(), gd.cuda())
    assert tuple(plain.shape) == (2, C, 16 * h, 16 * w)
    two_pass = _f32(nchw_view(ops.resize_nhwc(to_nhwc_bf16(plain), H, W, "bilinear")))
    rms = two_pass.pow(2).mean().sqrt().item()
    d = (fused - two_pass).abs()
    assert d.max().item() < 4e-2 * max(1.0, two_pass.abs().max().item()) and d.pow(2).mean().sqrt().item() < 4e-3 * max(1.0, rms)
    # fp32 oracle: the stack without its final fix-up is not exposed, so compare through the (linear) fix-up
    ref = F.interpolate(oups.jbu_stack(src, gd, w_sd, "upsampler."), (H, W), mode="bilinear", align_corners=True)
    conv = stack.fixup_proj[1]
    fx = fused.cuda()
    full = (fx + 0.1 * F.conv2d(fx, conv.weight.float(), conv.bias.float())).cpu()
    err = (full - ref).abs()
    assert err.max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 5e-3 * max(1.0, ref.pow(2).mean().sqrt().item())
    # a size that is not 7/8 of the map: forward_stages leaves the resize to the caller
    other = stack.forward_stages(src.cuda(), gd.cuda(), out_size=(H - 2, W))
    assert tuple(other.shape) == (2, C, 16 * h, 16 * w)


def test_jbu_composite_stage_vs_plain_fp32_stage():
    """The product path's composite-kernel / MFMA JBU stage against the stage-by-stage fp32 kernels of csrc/jbu_f32.hip
    (per-pixel 49-tap kernels, bicubic x2, adaptive 7x7 conv): two separate derivations of the published algorithm."""
    from isegprobe_amd import hip_ops as ops
    from isegprobe_amd.core.model.upsamplers import JBUFeatUpUpsampler
    torch.manual_seed(3)
    C = 128
    up = seeded_(JBUFeatUpUpsampler("dinov2", feat_dim=C), 9).cuda()
    st = up.upsampler.up2
    with torch.no_grad():
        st.range_temp.fill_(0.7)
        st.sigma_spatial.fill_(0.9)
    src = torch.randn(2, 12, 20, C, device="cuda")
    g = torch.randn(2, 3, 90, 150, device="cuda")
    f32 = lambda t: t.detach().float().contiguous()
    small = ops.adaptive_avg_pool(g, 24, 40)
    proj = ops.jbu_range_proj(small, f32(st.range_proj[0].weight.flatten(1)), f32(st.range_proj[0].bias),
                              f32(st.range_proj[3].weight.flatten(1)), f32(st.range_proj[3].bias), exact=True)
    ref = ops.jbu_stage_f32(src.to(torch.bfloat16).float(), proj, small, f32(st.fixup_proj[0].weight.flatten(1)),
                            f32(st.fixup_proj[0].bias), f32(st.fixup_proj[3].weight.flatten(1)), f32(st.fixup_proj[3].bias),
                            0.7, 0.9)
    y = st.run(src.to(torch.bfloat16), g).float()
    err = (y - ref).abs()
    assert err.max().item() < 3e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 5e-3 * max(1.0, ref.pow(2).mean().sqrt().item())


def test_attention_hd256():
    """head_dim 197 zero-padded to 256: LoftUp(n_dim=768)'s cross-attention (BASELINE configs[4])."""
    from isegprobe_amd import hip_ops as ops
    torch.manual_seed(2)
    B, Lq, Lk, H, hd, real = 2, 500, 260, 4, 256, 197
    def mk(L):
        t = torch.randn(B, L, H, hd, device="cuda")
        t[..., real:] = 0
        return t.to(torch.bfloat16)
    q, k, v = mk(Lq), mk(Lk), mk(Lk)
    out = ops.attention(q, k, v, real ** -0.5)
    p = ((q.float().permute(0, 2, 1, 3) * real ** -0.5) @ k.float().permute(0, 2, 3, 1)).softmax(-1)
    ref = (p @ v.float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert (out.float() - ref).abs().max().item() < 2e-2
    assert out[..., real:].abs().max().item() == 0


def test_attention_hd128():
    from isegprobe_amd import hip_ops as ops
    torch.manual_seed(3)
    B, Lq, Lk, H = 2, 300, 200, 4
    q = torch.randn(B, Lq, H, 128, device="cuda").to(torch.bfloat16)
    k = torch.randn(B, Lk, H, 128, device="cuda").to(torch.bfloat16)
    v = torch.randn(B, Lk, H, 128, device="cuda").to(torch.bfloat16)
    out = ops.attention(q, k, v, 101 ** -0.5)
    p = ((q.float().permute(0, 2, 1, 3) * 101 ** -0.5) @ k.float().permute(0, 2, 3, 1)).softmax(-1)
    ref = (p @ v.float().permute(0, 2, 1, 3)).permute(0, 2, 1, 3)
    assert (out.float() - ref).abs().max().item() < 2e-2


def test_loftup_fourier_minmax_kernels():
    from isegprobe_amd import hip_ops as ops
    from oracle import upsamplers as oups
    torch.manual_seed(5)
    img = torch.randn(2, 3, 20, 28)
    mm = ops.minmax_nchw(img.cuda()).cpu()
    assert torch.equal(mm[:, 0], img.amin(dim=(0, 2, 3))) and torch.equal(mm[:, 1], img.amax(dim=(0, 2, 3)))
    biases = torch.randn(2, 5, 20)
    gamma, beta = 1 + 0.2 * torch.randn(203), 0.2 * torch.randn(203)
    feats = oups._implicit_feats(oups._minmax(img), biases, 20, True)
    ref = oups._channel_ln(feats, gamma, beta)
    freqs = torch.exp(torch.linspace(-2, 10, 20))
    out = ops.loftup_fourier_cn(img.cuda(), mm.cuda(), freqs.cuda(), biases[0].reshape(-1).cuda().contiguous(),
                                biases[1].reshape(-1).cuda().contiguous(), gamma.cuda(), beta.cuda(), 256)
    got = out.float().cpu().permute(0, 3, 1, 2)
    assert got[:, 203:].abs().max().item() == 0
    # sin/cos at frequencies up to e^10: fp32 argument rounding alone is ~1e-3 rad; bf16 output
    assert (got[:, :203] - ref).abs().max().item() < 3e-2


def test_loftup_vs_golden(golden):
    from isegprobe_amd.core.model.upsamplers import LoftUpUpsampler
    g = golden("upsamplers_head")
    up = LoftUpUpsampler(None, n_dim=128)
    missing, unexpected = up.load_state_dict(weights_from(g, "loftup_w"), strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    src = torch.from_numpy(g["source"])[:, :, :2, :3].contiguous().cuda()
    y = _f32(up.cuda().eval()(src, torch.from_numpy(g["loftup_guidance"]).cuda()))
    ref = torch.from_numpy(g["loftup_y"])
    assert y.shape == ref.shape
    err = (y - ref).abs()
    print("loftup max err", err.max().item(), "rms", err.pow(2).mean().sqrt().item(), "ref rms", ref.pow(2).mean().sqrt().item())
    assert err.max().item() < 6e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 1e-2 * max(1.0, ref.pow(2).mean().sqrt().item())


def test_lift_vs_golden(golden):
    from isegprobe_amd.core.model.upsamplers import LiFTUpsampler
    g = golden("upsamplers_head")
    up = LiFTUpsampler(None, n_dim=128, patch=14)
    missing, unexpected = up.load_state_dict(weights_from(g, "lift_w"), strict=False)
    assert not unexpected and all("num_batches_tracked" in k for k in missing), (missing, unexpected)
    y = _f32(up.cuda().eval()(torch.from_numpy(g["source"]).cuda(), torch.from_numpy(g["guidance"]).cuda()))
    ref = torch.from_numpy(g["lift_y"])
    assert y.shape == ref.shape
    err = (y - ref).abs()
    print("lift max err", err.max().item(), "rms", err.pow(2).mean().sqrt().item(), "ref rms", ref.pow(2).mean().sqrt().item())
    assert err.max().item() < 3e-2 * max(1.0, ref.abs().max().item())


def test_loftup_768_vs_oracle():
    """LoftUp(n_dim=768), the upsampler of BASELINE configs[4] (DINOv2-B/14): 788 channels (832 padded), 4 heads of
    197 (padded to 256 for the fused attention), vs the CPU oracle with seeded weights; and its gradient w.r.t. the LR
    features (head_dim-256 attention backward) vs autograd of the oracle."""
    from isegprobe_amd.core.model.upsamplers import LoftUpUpsampler
    from oracle import upsamplers as oups
    torch.manual_seed(4)
    up = seeded_(LoftUpUpsampler(None, n_dim=768), 11)
    w = {"upsampler." + k: v.clone() for k, v in up.upsampler.state_dict().items()}
    src = torch.randn(1, 768, 3, 4)
    gd = torch.randn(1, 3, 42, 56)
    src_ref = src.clone().requires_grad_(True)
    ref = oups.loftup(src_ref, gd, w, "upsampler.")
    coef = torch.randn_like(ref)
    (ref * coef).sum().backward()
    up = up.cuda().eval()
    with torch.no_grad():
        y = _f32(up(src.cuda(), gd.cuda()))
    err = (y - ref.detach()).abs()
    print("loftup768 max err", err.max().item(), "rms", err.pow(2).mean().sqrt().item(), "ref rms", ref.pow(2).mean().sqrt().item())
    assert err.max().item() < 6e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 1e-2 * max(1.0, ref.pow(2).mean().sqrt().item())
    s = src.cuda().to(torch.bfloat16).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2).requires_grad_(True)  # NHWC storage
    out = up(s, gd.cuda())
    (out.float() * coef.cuda()).sum().backward()
    got, want = s.grad.float().cpu(), src_ref.grad
    cos = torch.nn.functional.cosine_similarity(got.flatten(), want.flatten(), dim=0).item()
    rms = (got - want).pow(2).mean().sqrt().item() / want.pow(2).mean().sqrt().item()
    print(f"loftup768 grad: cos {cos:.6f} rms-rel {rms:.3e}")
    assert cos > 0.995 and rms < 0.1


@pytest.mark.parametrize("n_dim", [44, 240])
def test_loftup_odd_widths_fall_back_to_bf16(n_dim):
    """Pixel-stream widths whose padded count (64 for n_dim 44, 320 for n_dim 240) does not tile into the f16 conv's 192- /
    128-channel blocks must take the bf16 stream at inference instead of raising (the half stream is selected per width)."""
    from isegprobe_amd import hip_ops as ops
    from isegprobe_amd.core.model.upsamplers import LoftUpUpsampler
    from oracle import upsamplers as oups
    torch.manual_seed(n_dim)
    up = seeded_(LoftUpUpsampler(None, n_dim=n_dim), 13)
    assert not ops.conv_takes_f16(up._cp())
    w = {"upsampler." + k: v.clone() for k, v in up.upsampler.state_dict().items()}
    src, gd = torch.randn(2, n_dim, 3, 2), torch.rand(2, 3, 42, 28)
    ref = oups.loftup(src, gd, w, "upsampler.")
    with torch.no_grad():
        y = _f32(up.cuda().eval()(src.cuda(), gd.cuda()))
    err = (y - ref).abs()
    print(f"loftup n_dim {n_dim}: max {err.max().item():.3g} rms {err.pow(2).mean().sqrt().item():.3g}")
    assert err.max().item() < 6e-2 * max(1.0, ref.abs().max().item())
    assert err.pow(2).mean().sqrt().item() < 1e-2 * max(1.0, ref.pow(2).mean().sqrt().item())


def test_frozen_checkpoint_files_vs_reference_outputs(golden):
    """The three frozen-weight files (tests/golden/frozen_ckpts/, layouts of the reference's loaders) loaded through the plugin
    constructors -- LoftUpUpsampler(upsampler_path=), LiFTUpsampler(lift_path=), DINOv2Featurizer(weights=) -- against the
    outputs the reference produced after reading the same files with its own loaders."""
    import os
    from conftest import GOLDEN
    from helpers import TINY_VIT
    from isegprobe_amd.core.model.featurizers import DINOv2Featurizer
    from isegprobe_amd.core.model.upsamplers import LiFTUpsampler, LoftUpUpsampler
    g = golden("frozen_ckpts")
    root = os.path.join(GOLDEN, "frozen_ckpts")
    src, gd = torch.from_numpy(g["source"]).cuda(), torch.from_numpy(g["guidance"]).cuda()
    for name, up in (("loftup", LoftUpUpsampler(os.path.join(root, "loftup_tiny.ckpt"), n_dim=64)),
                     ("lift", LiFTUpsampler(os.path.join(root, "lift_tiny.pth"), n_dim=64, patch=14))):
        with torch.no_grad():
            y = _f32(up.cuda().eval()(src, gd))
        ref = torch.from_numpy(g[name + "_y"])
        assert y.shape == ref.shape
        err = (y - ref).abs()
        print(name, "max err", err.max().item(), "rms", err.pow(2).mean().sqrt().item(), "ref rms", ref.pow(2).mean().sqrt().item())
        assert err.max().item() < (6e-2 if name == "loftup" else 3e-2) * max(1.0, ref.abs().max().item())
        assert err.pow(2).mean().sqrt().item() < 1e-2 * max(1.0, ref.pow(2).mean().sqrt().item())
    f = DINOv2Featurizer("custom", "before_backbone", weights=os.path.join(root, "dinov2_tiny_hub.pth"), vit_kwargs=TINY_VIT).cuda().eval()
    with torch.no_grad():
        y = f(torch.from_numpy(g["dino_x"]).cuda(), torch.from_numpy(g["dino_clicks"]).cuda()).float().cpu()
    ref = torch.from_numpy(g["dino_y"])
    assert y.shape == ref.shape and (y - ref).abs().max().item() < 1e-2 * max(1.0, ref.abs().max().item())
