"""MaskCLIP featurizer (reference core/model/featurizers/MaskCLIP.py:13-92 on CLIP's VisionTransformer,
maskclip/model.py:286-430): CLIP ViT with the MaskCLIP dense read-out -- every block but the last runs
normally (pre-LN, nn.MultiheadAttention, QuickGELU MLP), the last block contributes only
out_proj(v_proj(ln_1 x)) (model.py:251-263), then cls drop, ln_post and the 768->512 projection.

``CLIPVisual`` is a parameter container with CLIP's ``visual.*`` state-dict layout; the forward pass is
HIP launches on the shared ViT kernels (bf16 operands; the reference runs fp16 weights on the GPU).
Weights: ``weights=`` (a CLIP state dict / path; non-visual keys are ignored) or
``$ISEGPROBE_CLIP_WEIGHTS``; the reference downloads them (maskclip/clip.py:118-177)."""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import hip_ops as ops
from ...utils.log import logger
from .._autograd import MaskCLIPTrunkFn, TokenAddFn, TokenInjectFn
from .._tensor import BF16, PackedCache, nchw_view

CLIP_ARCHS = {"ViT-B/16": dict(input_resolution=224, patch_size=16, width=768, layers=12, heads=12, output_dim=512)}


def _pad64(k):
    return (k + 63) // 64 * 64


class _ResBlock(nn.Module):
    def __init__(self, d, heads):
        super().__init__()
        self.attn = nn.MultiheadAttention(d, heads)
        self.ln_1 = nn.LayerNorm(d)
        self.mlp = nn.Sequential()
        self.mlp.add_module("c_fc", nn.Linear(d, d * 4))
        self.mlp.add_module("gelu", nn.Identity())
        self.mlp.add_module("c_proj", nn.Linear(d * 4, d))
        self.ln_2 = nn.LayerNorm(d)


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.Sequential(*[_ResBlock(width, heads) for _ in range(layers)])


class CLIPVisual(nn.Module):
    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        if width // heads != 64:
            raise NotImplementedError("the fused attention kernel is built for head_dim 64")
        self.patch_size, self.width, self.heads, self.output_dim = patch_size, width, heads, output_dim
        scale = width ** -0.5
        self.conv1 = nn.Conv2d(3, width, kernel_size=patch_size, stride=patch_size, bias=False)
        self.class_embedding = nn.Parameter(scale * torch.randn(width))
        self.positional_embedding = nn.Parameter(scale * torch.randn((input_resolution // patch_size) ** 2 + 1, width))
        self.ln_pre = nn.LayerNorm(width)
        self.transformer = _Transformer(width, layers, heads)
        self.ln_post = nn.LayerNorm(width)
        self.proj = nn.Parameter(scale * torch.randn(width, output_dim))


class _ClipHolder(nn.Module):
    def __init__(self, visual):
        super().__init__()
        self.visual = visual


class MaskCLIPFeaturizer(nn.Module):
    def __init__(self, model_name: str = "ViT-B/16", feats_injection_mode: str = "no_injection", weights=None,
                 visual_kwargs=None) -> None:
        super().__init__()
        self.feats_injection_mode = feats_injection_mode
        if visual_kwargs is None:
            if model_name not in CLIP_ARCHS:
                raise ValueError(f"Currently unsupported model_name for MaskCLIP: {model_name}")
            visual_kwargs = CLIP_ARCHS[model_name]
        self.model = _ClipHolder(CLIPVisual(**visual_kwargs))
        weights = weights or os.environ.get("ISEGPROBE_CLIP_WEIGHTS")
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights
            self.model.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.")})
            logger.info(f"Loaded checkpoint for MaskCLIP: {model_name}")
        else:
            logger.info(f"MaskCLIP {model_name}: no weights given, keeping random init (no network for clip.load)")
        self.model.eval()
        self.patch_size = self.model.visual.patch_size
        self._packed = PackedCache()
        self._pos_cache = {}

    def packed(self):
        def build():
            v = self.model.visual
            f32 = lambda t: t.detach().float().contiguous()
            b16 = lambda t: t.detach().to(BF16).contiguous()
            K = v.conv1.weight[0].numel()
            wp = torch.zeros(v.width, _pad64(K), device=v.conv1.weight.device, dtype=BF16)
            wp[:, :K] = v.conv1.weight.detach().flatten(1).to(BF16)
            D = v.width
            blocks = []
            for blk in v.transformer.resblocks:
                blocks.append(dict(
                    l1w=f32(blk.ln_1.weight), l1b=f32(blk.ln_1.bias), l2w=f32(blk.ln_2.weight), l2b=f32(blk.ln_2.bias),
                    qkv_w=b16(blk.attn.in_proj_weight), qkv_b=f32(blk.attn.in_proj_bias),
                    v_w=b16(blk.attn.in_proj_weight[-D:]), v_b=f32(blk.attn.in_proj_bias[-D:]),
                    out_w=b16(blk.attn.out_proj.weight), out_b=f32(blk.attn.out_proj.bias),
                    fc_w=b16(blk.mlp.c_fc.weight), fc_b=f32(blk.mlp.c_fc.bias),
                    pj_w=b16(blk.mlp.c_proj.weight), pj_b=f32(blk.mlp.c_proj.bias)))
                d = blocks[-1]  # aliases in the key names the shared transformer-block backward expects
                d.update(n1w=d["l1w"], n1b=d["l1b"], n2w=d["l2w"], n2b=d["l2b"], proj_w=d["out_w"], proj_b=d["out_b"],
                         fc1_w=d["fc_w"], fc1_b=d["fc_b"], fc2_w=d["pj_w"], fc2_b=d["pj_b"], ls1=None, ls2=None)
            n_out = _pad64(v.output_dim)
            proj = torch.zeros(n_out, D, device=wp.device, dtype=BF16)
            proj[:v.output_dim] = v.proj.detach().t().to(BF16)
            self._pos_cache.clear()
            return dict(patch_w=wp, zero_b=torch.zeros(D, device=wp.device), blocks=blocks,
                        pre_w=f32(v.ln_pre.weight), pre_b=f32(v.ln_pre.bias), post_w=f32(v.ln_post.weight),
                        post_b=f32(v.ln_post.bias), proj=proj)
        return self._packed.get(list(self.model.parameters()), build)

    def _pos(self, rows, cols, H, W):
        """maskclip/interpolate.py:5-59 (bicubic, scale_factor (r+0.1)/M): [T+1, D] table + cls row, cached."""
        key = (rows, cols)
        if key not in self._pos_cache:
            with torch.no_grad():
                v = self.model.visual
                pe = v.positional_embedding.detach().float()
                N = pe.shape[0] - 1
                if not (rows * cols == N and H == W):
                    M = int(math.sqrt(N))
                    grid = F.interpolate(pe[1:].reshape(1, M, M, -1).permute(0, 3, 1, 2),
                                         scale_factor=((rows + 0.1) / M, (cols + 0.1) / M), mode="bicubic",
                                         align_corners=False, recompute_scale_factor=False)
                    assert grid.shape[-2:] == (rows, cols)
                    pe = torch.cat((pe[:1], grid.permute(0, 2, 3, 1).reshape(rows * cols, -1)), dim=0)
                table = pe.contiguous()
                self._pos_cache[key] = (table, (v.class_embedding.detach().float() + table[0]).contiguous())
        return self._pos_cache[key]

    def forward(self, x: torch.Tensor, additional_features: torch.Tensor = None) -> torch.Tensor:
        b, _, H, W = x.shape
        p = self.patch_size
        h, w = H // p, W // p
        v = self.model.visual
        D, heads, T = v.width, v.heads, h * w
        mode = self.feats_injection_mode
        before = additional_features is not None and mode == "before_backbone"
        after = additional_features is not None and mode == "after_backbone"
        wants_grad = torch.is_grad_enabled() and additional_features is not None and additional_features.requires_grad
        P = self.packed()
        if wants_grad and before:  # the reference's training mode (models/sbd/maskclip/patch-embed_noup.py:41)
            with torch.no_grad():
                table, cls_row = self._pos(w, h, H, W)  # (reference quirk: rows/cols swapped on this route, see below)
                A = ops.patchify(x.float().contiguous(), None, None, p, P["patch_w"].shape[1])
                xs = torch.empty(b * (T + 1), D, device=x.device, dtype=torch.float32)
                ops.gemm(A, P["patch_w"], ops._epilogue(ops._lib.EP_TOKENS_F32, xs, D, P["zero_b"], None, table, T))
                xs.view(b, T + 1, D)[:, 0].copy_(cls_row)
            if tuple(additional_features.shape) != (b, T, D):
                raise AssertionError(f"x.shape: {(b, T, D)}, additional_features.shape: {tuple(additional_features.shape)}")
            x0 = TokenInjectFn.apply(xs, additional_features, b, T)
            feats = MaskCLIPTrunkFn.apply(x0, P, heads, b, T, v.output_dim)
            return nchw_view(feats.view(b, h, w, v.output_dim))
        with torch.no_grad():
            # Reference quirk kept: the before_backbone route interpolates the pos-embed grid with rows and
            # columns swapped (model.py:389,402-404 vs :322,341); identical for square inputs.
            table, cls_row = self._pos(w, h, H, W) if before else self._pos(h, w, H, W)
            A = ops.patchify(x.float().contiguous(), None, None, p, P["patch_w"].shape[1])
            xs = torch.empty(b * (T + 1), D, device=x.device, dtype=torch.float32)
            ops.gemm(A, P["patch_w"], ops._epilogue(ops._lib.EP_TOKENS_F32, xs, D, P["zero_b"], None, table, T))
            xs.view(b, T + 1, D)[:, 0].copy_(cls_row)
            if before:  # MaskCLIP.py:51-65
                if tuple(additional_features.shape) != (b, T, D):
                    raise AssertionError(f"x.shape: {(b, T, D)}, additional_features.shape: {tuple(additional_features.shape)}")
                ops.token_add_(xs, additional_features, b, T, has_cls=True)
            xs = ops.layernorm(xs, P["pre_w"], P["pre_b"], 1e-5, out_dtype=torch.float32)  # ln_pre
            nb = len(P["blocks"])
            for i, blk in enumerate(P["blocks"]):
                a = ops.layernorm(xs, blk["l1w"], blk["l1b"], 1e-5)
                if i == nb - 1:  # forward_v: value path only, no residual
                    vin = ops.linear(a, blk["v_w"], blk["v_b"])
                    vout = ops.linear(vin, blk["out_w"], blk["out_b"])
                    break
                qkv = ops.linear(a, blk["qkv_w"], blk["qkv_b"])
                att = ops.attention_packed_qkv(qkv, b, T + 1, heads, 64 ** -0.5)
                ops.linear_residual_(xs, att, blk["out_w"], blk["out_b"], None)
                m = ops.linear(ops.layernorm(xs, blk["l2w"], blk["l2b"], 1e-5), blk["fc_w"], blk["fc_b"], "quick_gelu")
                ops.linear_residual_(xs, m, blk["pj_w"], blk["pj_b"], None)
            post = ops.layernorm(vout, P["post_w"], P["post_b"], 1e-5, group_out=T, skip=1, rows_out=b * T)  # cls drop + ln_post
            feats = ops.linear(post, P["proj"], None)[:, :v.output_dim].contiguous()
            if after and not wants_grad:  # MaskCLIP.py:75-83
                if tuple(additional_features.shape) != (b, T, v.output_dim):
                    raise AssertionError(f"features.shape: {(b, T, v.output_dim)}, additional_features.shape: "
                                         f"{tuple(additional_features.shape)}")
                ops.token_add_(feats, additional_features, b, T, has_cls=False)
        if after and wants_grad:
            feats = TokenAddFn.apply(feats.view(b, T, -1), additional_features)
        return nchw_view(feats.view(b, h, w, v.output_dim))
