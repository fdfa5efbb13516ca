"""Horizontal-flip test-time augmentation (reference core/inference/transforms/flip.py:12-45)."""
import torch

from .... import hip_ops as ops
from .base_transform import BaseTransform


class AddHorizontalFlip(BaseTransform):
    def transform(self, image_nd, clicks_lists):
        assert len(image_nd.shape) == 4
        image_nd = torch.cat([image_nd, torch.flip(image_nd, dims=[3])], dim=0)  # data movement only
        width = image_nd.shape[3]
        flipped = [[c.copy(coords=(c.coords[0], width - c.coords[1] - 1)) for c in clicks] for clicks in clicks_lists]
        return image_nd, clicks_lists + flipped

    def inv_transform(self, prob_map: torch.Tensor) -> torch.Tensor:
        """0.5 * (map + flip(map_of_mirrored)); BasePredictor fuses this with the sigmoid that follows
        it in the inverse chain (``fused_with_sigmoid``)."""
        assert len(prob_map.shape) == 4 and prob_map.shape[0] % 2 == 0
        n = prob_map.shape[0] // 2
        return 0.5 * (prob_map[:n] + torch.flip(prob_map[n:], dims=[3]))

    @staticmethod
    def fused_with_sigmoid(logits: torch.Tensor) -> torch.Tensor:
        return ops.fuse_flip_sigmoid(logits, with_flip=True)

    def get_state(self) -> None:
        return None

    def set_state(self, state) -> None:
        pass

    def reset(self) -> None:
        pass
