"""Transformations used during evaluation (reference core/inference/transforms/__init__.py)."""
from .base_transform import BaseTransform, SigmoidForPred
from .crops import Crops
from .flip import AddHorizontalFlip
from .limit_longest_side import LimitLongestSide
from .zoom_in import ZoomIn

__all__ = ["BaseTransform", "SigmoidForPred", "Crops", "AddHorizontalFlip", "LimitLongestSide", "ZoomIn"]
