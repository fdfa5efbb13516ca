"""Base class for segmentation heads (reference core/model/heads/base_head.py:8-18)."""
from abc import abstractmethod

import torch
import torch.nn as nn

from .... import hip_ops as ops
from ...._lib import IspError
from .._autograd import ClassifierFn, grad_mode
from .._tensor import BF16, PackedCache


class BaseClassifierHead(nn.Module):
    def __init__(self, in_channels: int, num_classes: int) -> None:
        super().__init__()
        self.in_channels = in_channels
        self.num_classes = num_classes
        self.classifier = nn.Conv2d(in_channels, num_classes, kernel_size=1)  # parameter holder
        self._cls_packed = PackedCache()

    def _cls_weights(self):
        if self.num_classes != 1:
            raise IspError("only num_classes == 1 (binary interactive segmentation) is built")
        return self._cls_packed.get(
            (self.classifier.weight, self.classifier.bias),
            lambda: (self.classifier.weight.detach().float().reshape(-1).contiguous(),
                     float(self.classifier.bias.detach().float().item())))

    def _classify(self, x_nhwc, post_relu=False):
        """1x1 conv C -> num_classes on an NHWC bf16 map -> [B, num_classes, H, W] f32.  ``post_relu``: x comes out of
        a conv+ReLU layer (lets the backward fuse that ReLU's mask); never set it for a raw feature map."""
        if grad_mode(self.classifier) or (torch.is_grad_enabled() and x_nhwc.requires_grad):
            self._cls_weights()  # validates num_classes
            return ClassifierFn.apply(x_nhwc, self.classifier.weight, self.classifier.bias, post_relu)
        w, b = self._cls_weights()
        B, H, W, _ = x_nhwc.shape
        return ops.classifier(x_nhwc, w, b).view(B, 1, H, W)

    @abstractmethod
    def forward(self, x):
        pass
