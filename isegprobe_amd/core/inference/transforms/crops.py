"""Sliding-window evaluation transform (behaviour of reference core/inference/transforms/crops.py:14-117, pinned by the
reference's own outputs in tests/golden/crops.npz).

forward   An image at least as large as the window in both directions becomes a BATCH of overlapping windows: along
          each axis the fewest windows that still overlap their neighbours by ``min_overlap`` of the window size,
          evenly spread, the last one flush with the border (``window_starts``).  Clicks are shifted into every
          window's coordinates (not filtered: a click outside a window keeps its out-of-range coordinates, which the
          click-map kernel ignores).  A smaller image passes through unchanged.
inverse   The per-window probability maps are summed back at their offsets and divided by the number of windows that
          cover each pixel.

Host glue off the per-click hot path (the reference package exports it; none of its predictors' default transform
lists uses it): tensor slicing / accumulation only, on whichever device the image lives."""
from typing import List, Optional, Tuple

import torch

from .base_transform import BaseTransform


def get_offsets(length: int, crop_size: int, min_overlap_ratio: float = 0.2) -> List[int]:
    """Window start positions along an axis of ``length`` pixels."""
    if length == crop_size:
        return [0]
    ratio = length / crop_size
    count = -int(-(ratio - min_overlap_ratio) // (1 - min_overlap_ratio))  # ceil
    overlap = int(crop_size * ((count - ratio) / (count - 1)))               # whole pixels shared by neighbours
    step, last = crop_size - overlap, length - crop_size
    return [min(i * step, last) for i in range(count)]


class Crops(BaseTransform):
    def __init__(self, crop_size: Tuple[int, int] = (320, 480), min_overlap: float = 0.2) -> None:
        super().__init__()
        self.crop_height, self.crop_width = crop_size
        self.min_overlap = min_overlap
        self.reset()

    def reset(self) -> None:
        self.x_offsets: Optional[List[int]] = None
        self.y_offsets: Optional[List[int]] = None
        self._counts: Optional[torch.Tensor] = None  # windows covering each pixel (None: the image passed through)

    def get_state(self) -> Tuple:
        return self.x_offsets, self.y_offsets, self._counts

    def set_state(self, state) -> None:
        self.x_offsets, self.y_offsets, self._counts = state

    def _windows(self):
        return [(dy, dx) for dy in self.y_offsets for dx in self.x_offsets]  # row-major, the batch order

    def transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        height, width = image_nd.shape[2:4]
        self._counts = None
        if height < self.crop_height or width < self.crop_width:
            return image_nd, clicks_lists
        self.x_offsets = get_offsets(width, self.crop_width, self.min_overlap)
        self.y_offsets = get_offsets(height, self.crop_height, self.min_overlap)
        ch, cw = self.crop_height, self.crop_width
        counts = torch.zeros(height, width, device=image_nd.device, dtype=torch.float32)
        crops = []
        for dy, dx in self._windows():
            counts[dy:dy + ch, dx:dx + cw] += 1
            crops.append(image_nd[:, :, dy:dy + ch, dx:dx + cw])
        self._counts = counts
        clicks = clicks_lists[0]
        shifted = [[c.copy(coords=(c.coords[0] - dy, c.coords[1] - dx)) for c in clicks] for dy, dx in self._windows()]
        return torch.cat(crops, dim=0), shifted

    def inv_transform(self, prob_map):
        if self._counts is None:
            return prob_map
        ch, cw = self.crop_height, self.crop_width
        total = torch.zeros(1, 1, *self._counts.shape, dtype=prob_map.dtype, device=prob_map.device)
        for i, (dy, dx) in enumerate(self._windows()):
            total[0, 0, dy:dy + ch, dx:dx + cw] += prob_map[i, 0]
        return total / self._counts
