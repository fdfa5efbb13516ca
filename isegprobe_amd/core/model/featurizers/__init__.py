"""Featurizer plugin API (reference core/model/featurizers/__init__.py:6-23)."""
from abc import ABC

from torch import nn


class BaseFeaturizer(ABC, nn.Module):
    """Base class for all featurizers."""

    def forward(self, x, additional_features=None, **kwargs):
        raise NotImplementedError


from .DINOv2 import DINOv2Featurizer  # noqa: E402
from .DINO import DINOFeaturizer  # noqa: E402
from .simple_ViT import SimpleViTFeaturizer  # noqa: E402
from .MaskCLIP import MaskCLIPFeaturizer  # noqa: E402

__all__ = ["BaseFeaturizer", "DINOv2Featurizer", "DINOFeaturizer", "SimpleViTFeaturizer", "MaskCLIPFeaturizer"]
