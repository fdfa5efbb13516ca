"""Zoom-in transform of the click loop (behaviour of reference core/inference/transforms/zoom_in.py:13-256, pinned by
the reference's own runs in tests/golden/inference.npz: per-click ROIs, clicks, masks and IoUs for skip_clicks -1 and 1).

What it does, per call of the predictor:

  forward   Once more than ``skip_clicks`` clicks exist, the network no longer sees the whole image but a window
            around the object: the bounding box of {previous mask} + {positive clicks}, grown by ``expansion_ratio``
            (at least ``min_crop_size``), clamped to the image.  The window in use is only REPLACED when there is none
            yet, when a positive click falls outside it, or when it overlaps the freshly proposed one by less than
            ``recompute_thresh_iou`` -- so consecutive clicks usually reuse the same crop (and the upsamplers' cached
            guidance-only work with it).  The crop is resized (bilinear, align_corners) to ``target_size`` and the
            clicks are mapped into its coordinates (fractional, not rounded).
  inverse   The probability map is resized back to the window and pasted into a zero map of the image size; that
            full-size map is kept on the host as the "previous mask" of the next proposal.

Host-side integer logic only decides the window; both resizes are HIP kernels."""
from typing import List, Optional, Tuple

import torch

from .... import hip_ops as ops
from ...utils.misc import clamp_bbox, expand_bbox, get_bbox_from_mask, get_bbox_iou
from .base_transform import BaseTransform

Roi = Tuple[int, int, int, int]  # (first row, last row, first col, last col), inclusive


def _resize(x: torch.Tensor, height: int, width: int) -> torch.Tensor:
    return ops.resize_bilinear_nchw_f32(x.float().contiguous(), height, width)


def _roi_size(roi: Roi) -> Tuple[int, int]:
    return roi[1] - roi[0] + 1, roi[3] - roi[2] + 1


def get_object_roi(pred_mask, clicks_list, expansion_ratio, min_crop_size) -> Roi:
    """Window proposal: bbox of the mask with every positive click switched on, expanded and clamped."""
    marked = pred_mask.copy()
    for click in clicks_list:
        if click.is_positive:
            marked[int(click.coords[0]), int(click.coords[1])] = 1
    rows, cols = marked.shape[:2]
    return clamp_bbox(expand_bbox(get_bbox_from_mask(marked), expansion_ratio, min_crop_size), 0, rows - 1, 0, cols - 1)


def get_roi_image_nd(image_nd: torch.Tensor, object_roi: Roi, target_size) -> torch.Tensor:
    """Crop ``object_roi`` and resize it: to ``target_size`` exactly when that is a (height, width) tuple (aspect ratio
    not kept), else so that the longer side becomes ``target_size``."""
    r0, r1, c0, c1 = object_roi
    height, width = _roi_size(object_roi)
    if isinstance(target_size, tuple):
        out_h, out_w = target_size
    else:
        scale = target_size / max(height, width)
        out_h, out_w = int(round(height * scale)), int(round(width * scale))
    with torch.no_grad():
        return _resize(image_nd[:, :, r0:r1 + 1, c0:c1 + 1], out_h, out_w)


def check_object_roi(object_roi: Roi, clicks_list) -> bool:
    """True while every positive click lies inside the window (its last row / column excluded)."""
    r0, r1, c0, c1 = object_roi
    return all(r0 <= c.coords[0] < r1 and c0 <= c.coords[1] < c1 for c in clicks_list if c.is_positive)


class ZoomIn(BaseTransform):
    def __init__(self, target_size=400, skip_clicks: int = 1, expansion_ratio: float = 1.4, min_crop_size: int = 200,
                 recompute_thresh_iou: float = 0.5, prob_thresh: float = 0.50) -> None:
        super().__init__()
        self.target_size, self.skip_clicks = target_size, skip_clicks
        self.expansion_ratio, self.min_crop_size = expansion_ratio, min_crop_size
        self.recompute_thresh_iou, self.prob_thresh = recompute_thresh_iou, prob_thresh
        self.reset()

    # ------------------------------------------------------------------ state
    def reset(self) -> None:
        self._input_image_shape = None
        self._object_roi: Optional[Roi] = None  # the window in use
        self._prev_probs = None                 # full-size probabilities of the previous click (numpy, host)
        self._roi_image = None
        self.image_changed = False
        self.applied_roi = None                 # the crop the last transform() applied (None: image passed through)

    def get_state(self) -> Tuple:
        crop = None if self._roi_image is None else self._roi_image.cpu()
        return self._input_image_shape, self._object_roi, self._prev_probs, crop, self.image_changed

    def set_state(self, state: Tuple) -> None:
        self._input_image_shape, self._object_roi, self._prev_probs, self._roi_image, self.image_changed = state

    def _previous_mask(self):
        if self._prev_probs is None:
            return None
        mask = (self._prev_probs > self.prob_thresh)[0, 0]
        return mask if mask.any() else None

    # ------------------------------------------------------------------ forward
    def transform(self, image_nd, clicks_lists):
        pairs = [self._transform(image_nd[b:b + 1], [clicks]) for b, clicks in enumerate(clicks_lists)]
        return torch.cat([img for img, _ in pairs], dim=0), [cl[0] for _, cl in pairs]

    def _propose(self, image_nd, clicks) -> Optional[Roi]:
        mask = self._previous_mask()
        if mask is not None:
            return get_object_roi(mask, clicks, self.expansion_ratio, self.min_crop_size)
        if self.skip_clicks >= 0:
            return None  # nothing segmented yet: keep looking at the whole image
        return 0, image_nd.shape[2] - 1, 0, image_nd.shape[3] - 1

    def _must_replace(self, proposed: Roi, clicks) -> bool:
        current = self._object_roi
        return (current is None or not check_object_roi(current, clicks)
                or get_bbox_iou(proposed, current) < self.recompute_thresh_iou)

    def _transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        clicks = clicks_lists[0]
        self.image_changed, self.applied_roi = False, None
        if len(clicks) <= self.skip_clicks:
            return image_nd, clicks_lists
        self._input_image_shape = image_nd.shape
        proposed = self._propose(image_nd, clicks)
        if proposed is None:
            return image_nd, clicks_lists
        if self._must_replace(proposed, clicks):
            self._object_roi, self.image_changed = proposed, True
        self._roi_image = get_roi_image_nd(image_nd, self._object_roi, self.target_size)
        self.applied_roi = tuple(int(v) for v in self._object_roi)
        return self._roi_image.to(image_nd.device), [self._transform_clicks(clicks)]

    def _transform_clicks(self, clicks_list):
        """Image coordinates -> crop coordinates: (p - window start) * crop size / window size, kept fractional."""
        if self._object_roi is None:
            return clicks_list
        r0, _, c0, _ = self._object_roi
        win_h, win_w = _roi_size(self._object_roi)
        crop_h, crop_w = self._roi_image.shape[2:]
        return [c.copy(coords=(crop_h * (c.coords[0] - r0) / win_h, crop_w * (c.coords[1] - c0) / win_w)) for c in clicks_list]

    # ------------------------------------------------------------------ inverse
    def inv_transform(self, prob_map):
        return torch.cat([self._inv_transform(prob_map[b:b + 1]) for b in range(prob_map.shape[0])], dim=0)

    def _inv_transform(self, prob_map):
        if self._object_roi is not None:
            assert prob_map.shape[0] == 1
            r0, r1, c0, c1 = self._object_roi
            window = _resize(prob_map, *_roi_size(self._object_roi))
            if self._prev_probs is None:
                prob_map = window
            else:
                prob_map = torch.zeros(*self._prev_probs.shape, device=window.device, dtype=window.dtype)
                prob_map[:, :, r0:r1 + 1, c0:c1 + 1] = window
        self._prev_probs = prob_map.cpu().numpy()
        return prob_map

    def check_possible_recalculation(self) -> bool:
        """With skip_clicks <= 0 and no window chosen yet: would the mask just predicted on the whole image justify
        zooming (proposal covering less than half of the image)?  The predictor then predicts this click again."""
        if self._prev_probs is None or self._object_roi is not None or self.skip_clicks > 0:
            return False
        mask = self._previous_mask()
        if mask is None:
            return False
        whole = (0, self._input_image_shape[2] - 1, 0, self._input_image_shape[3] - 1)
        return get_bbox_iou(get_object_roi(mask, [], self.expansion_ratio, self.min_crop_size), whole) < 0.50
