"""Parameter-free upsamplers (reference core/model/upsamplers/basic_upsamplers.py:8-42),
as HBM-bound NHWC bf16 resize kernels."""
from .... import hip_ops as ops
import torch

from .._autograd import ResizeBilinearFn
from .._tensor import nchw_view, to_nhwc_bf16
from . import BaseUpsampler


class IdentityUpsampler(BaseUpsampler):
    """Identity upsampler that does not change the input tensor."""

    def __init__(self):
        super().__init__()

    def forward(self, source, guidance):
        return source


class _Resize(BaseUpsampler):
    mode = None

    def __init__(self):
        super().__init__()

    def forward(self, source, guidance):
        _, _, h, w = guidance.shape
        x = to_nhwc_bf16(source)
        if torch.is_grad_enabled() and x.requires_grad:
            if self.mode != "bilinear":
                raise NotImplementedError(f"backward of the {self.mode} resize is not built")
            return nchw_view(ResizeBilinearFn.apply(x, h, w))
        return nchw_view(ops.resize_nhwc(x, h, w, self.mode))


class NearestUpsampler(_Resize):
    mode = "nearest"


class BilinearUpsampler(_Resize):
    mode = "bilinear"  # align_corners=True


class BicubicUpsampler(_Resize):
    mode = "bicubic"  # align_corners=False
