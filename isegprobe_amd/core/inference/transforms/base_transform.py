"""Transform protocol + SigmoidForPred (reference core/inference/transforms/base_transform.py:10-48)."""
from typing import Dict

import torch

from .... import hip_ops as ops


class BaseTransform(object):
    def __init__(self) -> None:
        self.image_changed = False

    def transform(self, image_nd, clicks_lists):
        raise NotImplementedError

    def inv_transform(self, prob_map):
        raise NotImplementedError

    def reset(self) -> None:
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError

    def set_state(self, state) -> None:
        raise NotImplementedError


class SigmoidForPred(BaseTransform):
    def transform(self, image_nd, clicks_lists):
        return image_nd, clicks_lists

    def inv_transform(self, prob_map: torch.Tensor) -> torch.Tensor:
        return ops.fuse_flip_sigmoid(prob_map, with_flip=False)

    def reset(self) -> None:
        pass

    def get_state(self) -> None:
        return None

    def set_state(self, state: Dict) -> None:
        pass
