"""Upsampler plugin API and registry (reference core/model/upsamplers/__init__.py:6-44)."""
from abc import ABC, abstractmethod

import torch.nn as nn


class BaseUpsampler(nn.Module, ABC):
    """Base class for upsampling modules: forward(source [B,C,h,w], guidance [B,3,H,W])."""

    @abstractmethod
    def forward(self, source, guidance):
        pass

    def _refuse_source_grad(self, source):
        """The learned upsamplers run raw HIP kernels with no backward: asking autograd to
        differentiate through them (training with clicks injected before them) must fail loudly
        instead of silently producing zero gradients."""
        import torch
        if torch.is_grad_enabled() and source.requires_grad:
            raise NotImplementedError(
                f"{type(self).__name__} has no backward on the HIP path: train with the identity/bilinear "
                "upsampler, or run this upsampler under torch.no_grad()")


from .basic_upsamplers import (  # noqa: E402
    BicubicUpsampler,
    BilinearUpsampler,
    IdentityUpsampler,
    NearestUpsampler,
)
from .JBUFeatUp import JBUFeatUpUpsampler  # noqa: E402
from .LiFT import LiFTUpsampler  # noqa: E402
from .LoftUp import LoftUpUpsampler  # noqa: E402

# used to load upsamplers from config
UPSAMPLER_REGISTRY = {
    "identity": IdentityUpsampler,
    "nearest": NearestUpsampler,
    "bilinear": BilinearUpsampler,
    "bicubic": BicubicUpsampler,
    "jbu_featup": JBUFeatUpUpsampler,
    "lift": LiFTUpsampler,
    "loftup": LoftUpUpsampler,
}

__all__ = [
    "BicubicUpsampler", "BilinearUpsampler", "IdentityUpsampler", "NearestUpsampler",
    "JBUFeatUpUpsampler", "LiFTUpsampler", "LoftUpUpsampler", "BaseUpsampler",
]
