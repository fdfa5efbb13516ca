"""Convolutional segmentation heads (reference core/model/heads/conv_heads.py:10-73).

The reference builds its layers from mmcv's ``ConvModule`` (conv + bias -> ReLU with mmcv
1.6.2 defaults); ``ConvModule`` below is an mmcv-free parameter container with the same
state-dict keys (``convs.{i}.conv.{weight,bias}``).  3x3 layers run as bf16 implicit-GEMM
convolutions with bias+ReLU fused, 1x1 layers as GEMMs; the final C->1 classifier is a
per-pixel dot product."""
import os

import torch
import torch.nn as nn

from .... import hip_ops as ops
from .._autograd import ClassifierFn, Conv1x1ReluFn, Conv3x3ReluClassifierFn, Conv3x3ReluFn, grad_mode
from .._tensor import BF16, PackedCache, to_nhwc_bf16
from .base_head import BaseClassifierHead


HEAD_F16 = os.environ.get("ISEGPROBE_HEAD_F16", "1") != "0"  # f16 operands for the inference convolutions (see forward)
CONV_OF_BILINEAR = os.environ.get("ISEGPROBE_CONV_OF_BILINEAR", "1") != "0"  # first conv through the bilinear resize (forward_of_bilinear)
CONV_OF_BILINEAR_TRAIN = os.environ.get("ISEGPROBE_CONV_OF_BILINEAR_TRAIN", "1") != "0"  # ... in training too (its adjoint)
HEAD_W_BITS = int(os.environ.get("ISEGPROBE_HEAD_W_BITS", "8"))  # significant bits kept in the half-format weights (8..11)
if not 8 <= HEAD_W_BITS <= 11:
    raise ValueError(f"ISEGPROBE_HEAD_W_BITS={HEAD_W_BITS}: the half-format head weights keep 8 (bf16-valued) to 11 (full half) bits")


def _head_weight(w, dtype):
    """Kernel-layout weight in ``dtype``.  The half-format copy keeps HEAD_W_BITS significant bits (default 8 = the bf16
    values, which half holds exactly): the socket is at its power limit during the head's convolutions and the multipliers'
    power follows the operands' trailing zero bits -- measured 13.03 ms per launch with full half weights, 12.81 ms with
    8-bit ones, at 0.4e-3 of logit rms; the activations keep all 11 bits (tools/conv_f16_power.py)."""
    w = w.detach().float()
    if dtype != ops.F16 or HEAD_W_BITS >= 11:
        return w.to(dtype).contiguous()
    if HEAD_W_BITS == 8:
        return w.to(BF16).to(ops.F16).contiguous()
    drop = 24 - HEAD_W_BITS  # round the fp32 value to nearest-even at that bit (one rounding), then store it in half (exact)
    bits = w.contiguous().view(torch.int32)
    bits = (bits + ((1 << (drop - 1)) - 1) + ((bits >> drop) & 1)) & ~((1 << drop) - 1)
    return bits.view(torch.float32).to(ops.F16).contiguous()


class ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.activate = nn.ReLU(inplace=True)
        self._packed, self._packed_f16 = PackedCache(), PackedCache()

    def packed(self, dtype=BF16):
        """Kernel-layout weights in ``dtype``: bf16, or IEEE half for the f16 form of the 3x3 conv (``takes_f16``)."""
        def build():
            w = self.conv.weight.detach()
            n = w.shape[0]
            # [N, C, kh, kw] -> [N, kh*kw*C] (tap-major, channel-minor: the implicit-GEMM K order)
            return (_head_weight(w.permute(0, 2, 3, 1).reshape(n, -1), dtype), self.conv.bias.detach().float().contiguous())
        cache = self._packed if dtype == BF16 else self._packed_f16
        return cache.get((self.conv.weight, self.conv.bias), build)

    def takes_f16(self):
        """The f16 conv kernel exists for 3x3 layers whose output channels tile into 192- or (nearly) 128-channel blocks
        (``ops.conv_takes_f16``: 384 / 768 / 1024-wide heads and the 128-wide fixture models)."""
        return self.conv.kernel_size == (3, 3) and ops.conv_takes_f16(self.conv.out_channels) and self.conv.in_channels % 64 == 0

    def run(self, x_nhwc):
        if grad_mode(self.conv) or (torch.is_grad_enabled() and x_nhwc.requires_grad):
            fn = Conv3x3ReluFn if self.conv.kernel_size == (3, 3) else Conv1x1ReluFn
            return fn.apply(x_nhwc, self.conv.weight, self.conv.bias)
        if x_nhwc.dtype == ops.F16 and not self.takes_f16():
            x_nhwc = x_nhwc.to(BF16)
        w, b = self.packed(x_nhwc.dtype)
        if self.conv.kernel_size == (3, 3):
            return ops.conv3x3(x_nhwc, w, b, "relu")
        B, H, W, C = x_nhwc.shape
        return ops.linear(x_nhwc.view(-1, C), w, b, "relu").view(B, H, W, -1)


class SimpleClassifierHead(BaseClassifierHead):
    """Single 1x1 conv layer."""

    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__(in_channels, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._classify(to_nhwc_bf16(x))  # (BaseClassifierHead._classify records autograd when training)


class _StackedHead(BaseClassifierHead):
    kernel_size, padding = 1, 0

    def __init__(self, in_channels: int, num_layers: int, num_classes: int) -> None:
        super().__init__(in_channels, num_classes)
        self.num_layers = num_layers
        self.convs = nn.Sequential(*[
            ConvModule(in_channels, in_channels, kernel_size=self.kernel_size, stride=1, padding=self.padding)
            for _ in range(num_layers)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        layers = list(self.convs)
        f16_ok = (HEAD_F16 and layers and all(l.takes_f16() for l in layers) and self.num_classes == 1
                  and not (grad_mode(self) or (torch.is_grad_enabled() and x.requires_grad)))
        y = to_nhwc_bf16(x, keep_f16=f16_ok)
        if f16_ok and y.dtype != ops.F16:
            # inference: the convolutions run on IEEE-half operands (weights and the hidden map keep three more
            # mantissa bits; a bf16 input converts exactly) -- one extra pass over a map that is not half already
            y = ops.to_f16(y)
        return self._tail(y, layers)

    def _tail(self, y, layers):
        """Remaining conv layers + classifier; a trailing 3x3 layer is fused with the 1x1 classifier
        (its output map is never stored)."""
        training = grad_mode(self) or (torch.is_grad_enabled() and y.requires_grad)
        if (not training and layers and layers[-1].conv.kernel_size == (3, 3) and self.num_classes == 1
                and (y.numel() // y.shape[-1]) % 4 == 0):
            for layer in layers[:-1]:
                y = layer.run(y)
            if y.dtype == ops.F16 and not layers[-1].takes_f16():
                y = y.to(BF16)
            w, b = layers[-1].packed(y.dtype)
            wc, bc = self._cls_weights()
            B, H, W, _ = y.shape
            return ops.conv3x3_relu_classifier(y, w, b, wc, bc).view(B, 1, H, W)
        if y.dtype == ops.F16:  # (the remaining routes are bf16)
            y = y.to(BF16)
        if training and layers and layers[-1].conv.kernel_size == (3, 3) and self.num_classes == 1:
            for layer in layers[:-1]:
                y = layer.run(y)
            last = layers[-1].conv  # last conv + classifier as one autograd node (no separate ReLU-mask pass)
            return Conv3x3ReluClassifierFn.apply(y, last.weight, last.bias, self.classifier.weight, self.classifier.bias)
        for layer in layers:
            y = layer.run(y)
        return self._classify(y, post_relu=len(layers) > 0)


    def forward_folded_affine(self, x, Wf, bf, alpha):
        """forward(z) for z = x + alpha*(Wf x + bf) (per pixel) without materialising z: the map is
        folded into the first 3x3 conv's weights; its constant part becomes a tap table applied in
        the conv epilogue (exact at the zero-padded borders)."""
        first = self.convs[0]
        if first.conv.kernel_size != (3, 3):
            raise ValueError("folding needs a 3x3 first layer")

        def build():
            w = first.conv.weight.detach().float()                       # [N, C, 3, 3]
            n, c = w.shape[:2]
            wt = w.permute(0, 2, 3, 1).reshape(n, 9, c)                  # [N, tap, C]
            folded = wt + alpha * torch.matmul(wt, Wf.float())           # W1_t (I + a Wf)
            taps = alpha * torch.matmul(wt, bf.float())                  # [N, 9]
            bias_full = first.conv.bias.detach().float() + taps.sum(1)
            return (_head_weight(folded.reshape(n, 9 * c), dtype), bias_full.contiguous(), taps.t().contiguous())
        # an IEEE-half map (the FeatUp-JBU stack's own precision) keeps that precision through the head when the f16
        # conv exists for this layer; anything else is bf16
        x = x.permute(0, 2, 3, 1).contiguous() if x.dtype == ops.F16 and first.takes_f16() else to_nhwc_bf16(x)
        dtype = x.dtype
        caches = self.__dict__.setdefault("_fold_packed", {})
        cache = caches.setdefault(dtype, PackedCache())
        wfold, bias_full, taps = cache.get((first.conv.weight, first.conv.bias, Wf, bf), build)
        y = ops.conv3x3_folded_affine(x, wfold, bias_full, taps)
        return self._tail(y, list(self.convs)[1:])


    def of_bilinear_geometry(self, C, h, w, H, W):
        """Whether ``forward_of_bilinear`` has a kernel for a [*, C, h, w] map resized to H x W: 3x3 first layer, channel
        blocks of 64, up-scaling by ~5.7 or more (``ISEGPROBE_CONV_OF_BILINEAR=0`` switches the route off)."""
        if not CONV_OF_BILINEAR or self.kernel_size != 3 or self.num_layers < 1:
            return False
        conv = self.convs[0].conv
        return (C == conv.in_channels and C % 64 == 0 and (h, w) != (H, W)
                and ops.conv3x3_of_bilinear_supported(h, w, H, W, conv.out_channels))

    def of_bilinear_applies(self, x, H, W):
        """... and the call is on a GPU tensor: inference, or training with a bf16 stream (``forward_of_bilinear`` then records
        ``Conv3x3OfBilinearReluFn``; ``ISEGPROBE_CONV_OF_BILINEAR_TRAIN=0`` keeps training on resize + conv)."""
        if not x.is_cuda:
            return False
        if grad_mode(self) or (torch.is_grad_enabled() and x.requires_grad):
            if not CONV_OF_BILINEAR_TRAIN or x.dtype != BF16:
                return False
        return self.of_bilinear_geometry(x.shape[1], x.shape[2], x.shape[3], H, W)

    def forward_of_bilinear(self, x, H, W):
        """forward(F.interpolate(x, (H, W), mode="bilinear", align_corners=True)) without the resized map
        (iseg_probe_model.py:120-129 / basic_upsamplers.py:28-33 followed by conv_heads.py:69-73): the first convolution is
        linear in the resized map, so it runs as Z = x [W_0 .. W_8]^T at LOW resolution (one GEMM, 9*N columns) followed by
        the bilinear blend of the nine tap planes (csrc/conv_bilinear.hip) -- 36 multiply-adds per output value instead
        of 9*C, and the [B,H,W,C] map is never written.  x: [B,C,h,w]-shaped NHWC view (bf16 or half)."""
        first = self.convs[0]
        layers = list(self.convs)[1:]
        if grad_mode(self) or (torch.is_grad_enabled() and x.requires_grad):
            from .._autograd import Conv3x3OfBilinearReluFn
            y = Conv3x3OfBilinearReluFn.apply(to_nhwc_bf16(x), first.conv.weight, first.conv.bias, H, W)
            return self._tail(y, layers)
        xl = to_nhwc_bf16(x, keep_f16=True)
        if xl.dtype != ops.F16:
            xl = ops.to_f16(xl)  # bf16 values are exact in half
        B, h, w, C = xl.shape
        N = first.conv.out_channels

        def build():
            wt = first.conv.weight.detach().float()  # [N, C, 3, 3] -> rows t*N + n
            return (wt.permute(2, 3, 0, 1).reshape(9 * N, C).to(ops.F16).contiguous(), first.conv.bias.detach().float().contiguous())
        cache = self.__dict__.setdefault("_of_bilinear_packed", PackedCache())
        wz, bias = cache.get((first.conv.weight, first.conv.bias), build)
        z = ops.linear(xl.view(B * h * w, C), wz)
        half_next = HEAD_F16 and self.num_classes == 1 and layers and all(l.takes_f16() for l in layers)
        y = ops.conv3x3_of_bilinear_blend(z, bias, B, h, w, H, W, N, relu=True, out_dtype=ops.F16 if half_next else BF16)
        return self._tail(y, layers)


class SimpleConvSegHead(_StackedHead):
    """Several 1x1 conv layers."""


class ConvSegHead(_StackedHead):
    """Several 3x3 conv layers, followed by a 1x1 conv layer."""
    kernel_size, padding = 3, 1
