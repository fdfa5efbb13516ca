"""FeatUp JBU upsampler (reference core/model/upsamplers/JBUFeatUp.py:9-32), x16.

The reference obtains the module with ``torch.hub.load("mhamilton723/FeatUp", backbone_type,
use_norm=...).upsampler`` -- a third-party ``JBUStack`` whose source is not part of the
reference tree and cannot be fetched here.  ``JBUStack`` / ``JBULearnedRange`` below are
parameter containers with FeatUp's state-dict layout (``up{1..4}.{range_temp, range_proj.{0,3},
fixup_proj.{0,3}, sigma_spatial}``, ``fixup_proj.1``); the arithmetic is the HIP stage kernels
of csrc/jbu.hip (composite-kernel formulation, see its header).  Weights: ``weights=`` (state dict / path in FeatUp's upsampler key layout)
or ``$ISEGPROBE_JBU_WEIGHTS``; otherwise FeatUp's default init is kept.
"""
import os

import torch
import torch.nn as nn

from .... import hip_ops as ops
from ...utils.log import logger
from .._tensor import BF16, PackedCache, nchw_view, to_nhwc_bf16, pack_serial
from .._guidance_cache import GuidanceCache
from . import BaseUpsampler

FEAT_DIMS = {"dinov2": 384, "dino16": 384, "vit": 384, "maskclip": 512, "clip": 512, "resnet50": 2048}


class JBULearnedRange(nn.Module):
    def __init__(self, guidance_dim, feat_dim, key_dim, scale=2, radius=3):
        super().__init__()
        if (guidance_dim, key_dim, scale, radius) != (3, 32, 2, 3):
            raise NotImplementedError("the JBU kernels are built for guidance 3, key 32, x2, radius 3")
        d2 = (2 * radius + 1) ** 2
        self.range_temp = nn.Parameter(torch.tensor(0.0))
        self.range_proj = nn.Sequential(nn.Conv2d(guidance_dim, key_dim, 1, 1), nn.GELU(), nn.Dropout2d(0.1),
                                        nn.Conv2d(key_dim, key_dim, 1, 1))
        self.fixup_proj = nn.Sequential(nn.Conv2d(guidance_dim + d2, d2, 1, 1), nn.GELU(), nn.Dropout2d(0.1),
                                        nn.Conv2d(d2, d2, 1, 1))
        self.sigma_spatial = nn.Parameter(torch.tensor(1.0))
        self._packed = PackedCache()
        self._gcache = GuidanceCache()

    def packed(self):
        def build():
            f = lambda t: t.detach().float().contiguous()

            def pad64(w, b):  # fix-up MLP layer -> zero-padded [64,64] f16 + [64] f32 (MFMA operands)
                wp = torch.zeros(64, 64, device=w.device)
                wp[:w.shape[0], :w.shape[1]] = w.detach().float().flatten(1)
                bp = torch.zeros(64, device=w.device)
                bp[:b.shape[0]] = b.detach().float()
                return wp.to(ops.F16).contiguous(), bp
            f0w, f0b = pad64(self.fixup_proj[0].weight, self.fixup_proj[0].bias)
            f3w, f3b = pad64(self.fixup_proj[3].weight, self.fixup_proj[3].bias)
            return dict(w0=f(self.range_proj[0].weight.flatten(1)), b0=f(self.range_proj[0].bias),
                        w3=f(self.range_proj[3].weight.flatten(1)), b3=f(self.range_proj[3].bias),
                        f0w=f0w, f0b=f0b, f3w=f3w, f3b=f3b,
                        temp=float(self.range_temp.item()), sigma=float(self.sigma_spatial.item()))
        return self._packed.get(self._packed.tensors_of(self.parameters), build)

    def kernels(self, guidance, GH, GW, drops=None):
        """Composite (bicubic-x2 o 7x7) kernels of this stage: a function of the guidance only, so the click loop
        reuses them while the image / zoom-in ROI is unchanged (_guidance_cache)."""
        P = self.packed()
        if drops is not None:  # train-mode Dropout2d: fresh multipliers per forward, nothing to cache
            return self._build_kernels(P, guidance, GH, GW, drops)
        return self._gcache.get(guidance, pack_serial(P), (GH, GW), lambda: self._build_kernels(P, guidance, GH, GW))

    @staticmethod
    def _build_kernels(P, guidance, GH, GW, drops=None):
        small = ops.adaptive_avg_pool(guidance, GH, GW)
        proj = ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"], drop=None if drops is None else drops[0])
        return ops.jbu_kernels(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"],
                               drop=None if drops is None else drops[1])

    def run(self, source_nhwc, guidance, kc=None, out_dtype=None, drops=None):
        """One x2 stage.  Maps inside the stack are f16 (``out_dtype`` default); the stage that leaves it writes bf16."""
        # composite kernels on the low-res grid, applied by MFMA: no x2 map in HBM
        if kc is None:
            kc = self.kernels(guidance, source_nhwc.shape[1] * 2, source_nhwc.shape[2] * 2, drops)
        return ops.jbu_apply(source_nhwc, kc, out_dtype or ops.F16)

    @staticmethod
    def resize_fusable(GH, GW, OH, OW):
        """The stage's x2 map (GH x GW) resized to OH x OW can be produced in one pass when 8 stage rows / columns map
        onto exactly 7 output ones (FeatUp's x16 map vs. a patch-14 image: 512 -> 448)."""
        return GH % 8 == 0 and GW % 8 == 0 and OH * 8 == GH * 7 and OW * 8 == GW * 7

    def kernels_resized(self, guidance, GH, GW, OH, OW):
        """Kernel records of this stage blended onto the OH x OW grid of the model's bilinear resize (guidance-only,
        cached like ``kernels``); the stage's own records are never stored: the kernel blends them on the way out."""
        P = self.packed()

        def build():
            small = ops.adaptive_avg_pool(guidance, GH, GW)
            proj = ops.jbu_range_proj(small, P["w0"], P["b0"], P["w3"], P["b3"])
            return ops.jbu_kernels_resized(proj, small, P["f0w"], P["f0b"], P["f3w"], P["f3b"], P["temp"], P["sigma"], OH, OW)
        return self._gcache.get(guidance, pack_serial(P), (GH, GW, OH, OW), build)

    def run_resized(self, source_nhwc, guidance, OH, OW, kc9=None, out_dtype=BF16):
        """resize_bilinear(run(source), OH, OW) as one operator (isp_jbu_apply_resized on the blended records)."""
        if kc9 is None:
            kc9 = self.kernels_resized(guidance, source_nhwc.shape[1] * 2, source_nhwc.shape[2] * 2, OH, OW)
        return ops.jbu_apply_resized(source_nhwc, kc9, out_dtype=out_dtype)


class JBUStack(nn.Module):
    def __init__(self, feat_dim):
        super().__init__()
        self.up1 = JBULearnedRange(3, feat_dim, 32, radius=3)
        self.up2 = JBULearnedRange(3, feat_dim, 32, radius=3)
        self.up3 = JBULearnedRange(3, feat_dim, 32, radius=3)
        self.up4 = JBULearnedRange(3, feat_dim, 32, radius=3)
        self.fixup_proj = nn.Sequential(nn.Dropout2d(0.2), nn.Conv2d(feat_dim, feat_dim, kernel_size=1))
        self._packed = PackedCache()

    def forward_stages(self, source, guidance, out_size=None, records=None, out_dtype=BF16):
        """The four x2 stages WITHOUT the final fix-up  x + 0.1*conv1x1(x).  The fix-up is a per-pixel
        affine map; iSegProbeModel folds it (through the linear resize) into the seg head's first conv.
        With ``out_size`` the model's bilinear resize to the image size (iseg_probe_model.py:120-129) is fused into the
        last stage when the sizes allow it (otherwise the caller resizes as before).  ``out_dtype``: bf16, or IEEE half for
        a consumer that takes the stack's own precision (the seg head's f16 convolutions)."""
        x = to_nhwc_bf16(source, keep_f16=True)  # (the stack converts to half anyway)
        if records is None:
            records = self.stage_records(guidance, x.shape[1], x.shape[2], out_size)
        for up, kc in zip((self.up1, self.up2, self.up3), records[:3]):
            x = up.run(x, None, kc)
        if records[3].shape[3] == 9:  # records of the resized grid
            return nchw_view(self.up4.run_resized(x, None, records[3].shape[1], records[3].shape[2], records[3],
                                                  out_dtype=out_dtype))
        return nchw_view(self.up4.run(x, None, records[3], out_dtype=out_dtype))

    def stage_records(self, guidance, h, w, out_size=None):
        """The kernel records of the four stages for an h x w source: functions of the guidance only, so they can be
        computed before (or, on another stream, while) the featurizer runs."""
        guidance = guidance.float().contiguous()
        recs = [up.kernels(guidance, h << (i + 1), w << (i + 1)) for i, up in enumerate((self.up1, self.up2, self.up3))]
        GH, GW = h << 4, w << 4
        if out_size is not None and JBULearnedRange.resize_fusable(GH, GW, int(out_size[0]), int(out_size[1])):
            recs.append(self.up4.kernels_resized(guidance, GH, GW, int(out_size[0]), int(out_size[1])))
        else:
            recs.append(self.up4.kernels(guidance, GH, GW))
        return recs

    def dropout_active(self):
        """The reference trains with net.train() on the whole model (trainer.py:214), which also switches on the frozen
        stack's Dropout2d layers: 0.1 behind the GELU of every stage's range_proj and fixup_proj, 0.2 in front of the final
        1x1 conv.  Active whenever the module is in train mode, with or without autograd -- nn.Dropout2d's own rule (the
        reference's trainer calls net.eval() around its no-grad click-simulation forwards, trainer.py:404-414)."""
        return self.training

    def draw_dropout(self, B, device, generator=None):
        """One forward's Dropout2d multipliers (0 or 1/(1-p) per (image, channel)):
        {"stages": [(range [B,32], fixup [B,64], units 49.. are padding)] * 4, "fixup": [B,C]}."""
        def mult(n, p):
            keep = torch.rand(B, n, device=device, generator=generator) >= p
            return keep.float() / (1.0 - p)
        stages = []
        for _ in range(4):
            fix = torch.ones(B, 64, device=device)
            fix[:, :49] = mult(49, 0.1)
            stages.append((mult(32, 0.1), fix))
        return {"stages": stages, "fixup": mult(self.fixup_proj[1].in_channels, 0.2)}

    def fixup_affine(self):
        """(W [C,C], b [C], alpha): z = x + alpha * (W x + b)."""
        conv = self.fixup_proj[1]
        return conv.weight.detach().flatten(1), conv.bias.detach(), 0.1

    def forward(self, source, guidance, drops=None):
        """``drops``: train-mode Dropout2d multipliers (``draw_dropout``); drawn here when ``dropout_active()``."""
        x = to_nhwc_bf16(source, keep_f16=True)  # (the stack converts to half anyway)
        guidance = guidance.float().contiguous()
        if drops is None and self.dropout_active():
            drops = getattr(self, "fixed_dropout", None) or self.draw_dropout(x.shape[0], x.device)
        st = [None] * 4 if drops is None else drops["stages"]
        for up, d in zip((self.up1, self.up2, self.up3), st):
            x = up.run(x, guidance, drops=d)
        x = self.up4.run(x, guidance, out_dtype=BF16, drops=st[3])
        conv = self.fixup_proj[1]
        w, b = self._packed.get((conv.weight, conv.bias),
                                lambda: (conv.weight.detach().flatten(1).to(BF16).contiguous(),
                                         conv.bias.detach().float().contiguous()))
        B, H, W, C = x.shape
        xin = x if drops is None else (x * drops["fixup"].to(BF16).view(B, 1, 1, C)).contiguous()
        y = ops.linear_axpy_res(xin.view(-1, C), w, b, x.view(-1, C), 0.1)
        return nchw_view(y.view(B, H, W, C))


class JBUFeatUpUpsampler(BaseUpsampler):
    """Learned JBU upsampler from FeatUp. Performs x16 upsampling."""

    def __init__(self, backbone_type: str = None, use_norm: bool = True, weights=None, feat_dim: int = None) -> None:
        super().__init__()
        self.backbone_type = backbone_type
        self.use_norm = use_norm
        assert self.backbone_type in FEAT_DIMS, f"Invalid model type: {self.backbone_type}"
        self.upsampler = JBUStack(feat_dim or FEAT_DIMS[backbone_type])
        weights = weights or os.environ.get("ISEGPROBE_JBU_WEIGHTS")
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights
            self.upsampler.load_state_dict(sd)
        else:
            logger.info("JBUFeatUpUpsampler: no weights given, keeping default init (no network for torch.hub)")
        self.eval()

    def forward(self, source: torch.Tensor, guidance: torch.Tensor, drops=None) -> torch.Tensor:
        """``drops``: explicit train-mode Dropout2d multipliers (tests); by default the stack draws its own whenever the
        module is in train mode and autograd records (``JBUStack.dropout_active``)."""
        x = to_nhwc_bf16(source, keep_f16=True)  # (the stack converts to half anyway)
        if torch.is_grad_enabled() and x.requires_grad:  # training with clicks injected before the upsampler
            stack = self.upsampler
            if drops is None and stack.dropout_active():
                drops = getattr(stack, "fixed_dropout", None) or stack.draw_dropout(x.shape[0], x.device)  # (tests pin them)
            return nchw_view(_JBUFn.apply(x, guidance, stack, drops))
        return self.upsampler(source, guidance, drops=drops)


class _JBUFn(torch.autograd.Function):
    """JBUStack as one autograd node.  Every stage is LINEAR in the source (its composite kernels depend on the
    guidance only) and so is the fix-up z = x + 0.1 (W x + b): the backward is the adjoint chain
    g <- g + 0.1 W^T g, then four adjoint applies with the saved kernels."""

    @staticmethod
    def forward(ctx, src, guidance, stack, drops=None):
        x = src.detach()
        g = guidance.detach().float().contiguous()
        kcs = []  # (drops are drawn by the caller: grad mode is off inside a Function's forward)
        for i, up in enumerate((stack.up1, stack.up2, stack.up3, stack.up4)):
            kc = up.kernels(g, x.shape[1] * 2, x.shape[2] * 2, None if drops is None else drops["stages"][i])
            kcs.append(kc)
            x = ops.jbu_apply(x, kc, BF16 if i == 3 else ops.F16)
        conv = stack.fixup_proj[1]
        w32 = conv.weight.detach().flatten(1).float()
        bias = conv.bias.detach().float().contiguous()
        B, H, W, C = x.shape
        if drops is None:
            y = ops.linear_axpy_res(x.view(-1, C), w32.to(BF16).contiguous(), bias, x.view(-1, C), 0.1).view(B, H, W, C)
            ctx.wT = w32.t().contiguous().to(BF16)
        else:
            # Dropout2d(0.2) in front of the 1x1 conv: z = x + 0.1 (W (m o x) + b) with one multiplier per (image, channel), i.e.
            # image b sees the weights W diag(m_b) -- folded into B small weight matrices, one GEMM per image, instead of
            # masking (and, in the backward, converting / scaling / adding) the full-resolution map with framework kernels
            m = drops["fixup"].float().view(B, 1, C)
            wb = (w32[None] * m).to(BF16).contiguous()              # [B, C_out, C_in]: W diag(m_b)
            y = torch.empty_like(x)
            for bi in range(B):
                ops.linear_axpy_res(x[bi].view(-1, C), wb[bi], bias, x[bi].view(-1, C), 0.1, out=y[bi].view(-1, C))
            ctx.wT = wb.transpose(1, 2).contiguous()                # [B, C_in, C_out]: (W diag(m_b))^T
        ctx.kcs, ctx.per_image = kcs, drops is not None
        return y

    @staticmethod
    def backward(ctx, g_out):
        B, H, W, C = g_out.shape
        g = g_out.contiguous()
        if not ctx.per_image:
            g = ops.linear_axpy_res(g.view(-1, C), ctx.wT, None, g.view(-1, C), 0.1).view(B, H, W, C)  # (I + 0.1 W)^T
        else:  # (I + 0.1 W diag(m_b))^T g_b, image by image
            gi = torch.empty_like(g)
            for bi in range(B):
                ops.linear_axpy_res(g[bi].view(-1, C), ctx.wT[bi], None, g[bi].view(-1, C), 0.1, out=gi[bi].view(-1, C))
            g = gi
        for kc in reversed(ctx.kcs):
            g = ops.jbu_apply_bwd(g, kc)
        ctx.kcs = None
        return g, None, None, None
