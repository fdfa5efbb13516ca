"""Zoom-in on the object ROI (reference core/inference/transforms/zoom_in.py:13-256): crop the
ROI of the previous prediction, resize it to the network size (bilinear, align_corners=True),
rescale the clicks, and paste the prediction back.  Host ROI logic as in the reference; the
resizes are HIP kernels."""
from typing import List, Tuple

import torch

from .... import hip_ops as ops
from ...utils.misc import clamp_bbox, expand_bbox, get_bbox_from_mask, get_bbox_iou
from .base_transform import BaseTransform


class ZoomIn(BaseTransform):
    def __init__(self, target_size=400, skip_clicks: int = 1, expansion_ratio: float = 1.4,
                 min_crop_size: int = 200, recompute_thresh_iou: float = 0.5, prob_thresh: float = 0.50) -> None:
        super().__init__()
        self.target_size = target_size
        self.min_crop_size = min_crop_size
        self.skip_clicks = skip_clicks
        self.expansion_ratio = expansion_ratio
        self.recompute_thresh_iou = recompute_thresh_iou
        self.prob_thresh = prob_thresh
        self.reset()

    def transform(self, image_nd, clicks_lists):
        images, clicks = [], []
        for b in range(len(clicks_lists)):
            img, cl = self._transform(image_nd[b].unsqueeze(0), [clicks_lists[b]])
            images.append(img)
            clicks.append(cl[0])
        return torch.cat(images, dim=0), clicks

    def _transform(self, image_nd, clicks_lists):
        assert image_nd.shape[0] == 1 and len(clicks_lists) == 1
        self.image_changed = False
        self.applied_roi = None  # the crop this call actually applied (None: image passed through)
        clicks_list = clicks_lists[0]
        if len(clicks_list) <= self.skip_clicks:
            return image_nd, clicks_lists
        self._input_image_shape = image_nd.shape

        current_object_roi = None
        if self._prev_probs is not None:
            current_pred_mask = (self._prev_probs > self.prob_thresh)[0, 0]
            if current_pred_mask.sum() > 0:
                current_object_roi = get_object_roi(current_pred_mask, clicks_list, self.expansion_ratio,
                                                    self.min_crop_size)
        if current_object_roi is None:
            if self.skip_clicks >= 0:
                return image_nd, clicks_lists
            current_object_roi = 0, image_nd.shape[2] - 1, 0, image_nd.shape[3] - 1

        if (self._object_roi is None or not check_object_roi(self._object_roi, clicks_list)
                or get_bbox_iou(current_object_roi, self._object_roi) < self.recompute_thresh_iou):
            self._object_roi = current_object_roi
            self.image_changed = True
        self._roi_image = get_roi_image_nd(image_nd, self._object_roi, self.target_size)
        self.applied_roi = tuple(int(v) for v in self._object_roi)
        return self._roi_image.to(image_nd.device), [self._transform_clicks(clicks_list)]

    def inv_transform(self, prob_map):
        return torch.cat([self._inv_transform(prob_map[b].unsqueeze(0)) for b in range(prob_map.shape[0])], dim=0)

    def _inv_transform(self, prob_map):
        if self._object_roi is None:
            self._prev_probs = prob_map.cpu().numpy()
            return prob_map
        assert prob_map.shape[0] == 1
        rmin, rmax, cmin, cmax = self._object_roi
        prob_map = ops.resize_bilinear_nchw_f32(prob_map.float().contiguous(), rmax - rmin + 1, cmax - cmin + 1)
        if self._prev_probs is not None:
            new_prob_map = torch.zeros(*self._prev_probs.shape, device=prob_map.device, dtype=prob_map.dtype)
            new_prob_map[:, :, rmin:rmax + 1, cmin:cmax + 1] = prob_map
        else:
            new_prob_map = prob_map
        self._prev_probs = new_prob_map.cpu().numpy()
        return new_prob_map

    def check_possible_recalculation(self) -> bool:
        if self._prev_probs is None or self._object_roi is not None or self.skip_clicks > 0:
            return False
        pred_mask = (self._prev_probs > self.prob_thresh)[0, 0]
        if pred_mask.sum() > 0:
            possible_object_roi = get_object_roi(pred_mask, [], self.expansion_ratio, self.min_crop_size)
            image_roi = (0, self._input_image_shape[2] - 1, 0, self._input_image_shape[3] - 1)
            if get_bbox_iou(possible_object_roi, image_roi) < 0.50:
                return True
        return False

    def get_state(self) -> Tuple:
        roi_image = self._roi_image.cpu() if self._roi_image is not None else None
        return self._input_image_shape, self._object_roi, self._prev_probs, roi_image, self.image_changed

    def set_state(self, state: Tuple) -> None:
        self._input_image_shape, self._object_roi, self._prev_probs, self._roi_image, self.image_changed = state

    def reset(self) -> None:
        self._input_image_shape = None
        self._object_roi = None   # (rmin, rmax, cmin, cmax)
        self._prev_probs = None   # previous prediction, numpy on the host
        self._roi_image = None
        self.image_changed = False

    def _transform_clicks(self, clicks_list):
        if self._object_roi is None:
            return clicks_list
        rmin, rmax, cmin, cmax = self._object_roi
        crop_height, crop_width = self._roi_image.shape[2:]
        out = []
        for click in clicks_list:  # fractional coordinates are kept (zoom_in.py:181-193)
            new_r = crop_height * (click.coords[0] - rmin) / (rmax - rmin + 1)
            new_c = crop_width * (click.coords[1] - cmin) / (cmax - cmin + 1)
            out.append(click.copy(coords=(new_r, new_c)))
        return out


def get_object_roi(pred_mask, clicks_list, expansion_ratio, min_crop_size):
    pred_mask = pred_mask.copy()
    for click in clicks_list:
        if click.is_positive:
            pred_mask[int(click.coords[0]), int(click.coords[1])] = 1
    bbox = expand_bbox(get_bbox_from_mask(pred_mask), expansion_ratio, min_crop_size)
    h, w = pred_mask.shape[0], pred_mask.shape[1]
    return clamp_bbox(bbox, 0, h - 1, 0, w - 1)


def get_roi_image_nd(image_nd, object_roi, target_size):
    rmin, rmax, cmin, cmax = object_roi
    height, width = rmax - rmin + 1, cmax - cmin + 1
    if isinstance(target_size, tuple):
        new_height, new_width = target_size
    else:
        scale = target_size / max(height, width)
        new_height, new_width = int(round(height * scale)), int(round(width * scale))
    with torch.no_grad():
        roi = image_nd[:, :, rmin:rmax + 1, cmin:cmax + 1].float().contiguous()
        return ops.resize_bilinear_nchw_f32(roi, new_height, new_width)


def check_object_roi(object_roi, clicks_list) -> bool:
    for click in clicks_list:
        if click.is_positive:
            if click.coords[0] < object_roi[0] or click.coords[0] >= object_roi[1]:
                return False
            if click.coords[1] < object_roi[2] or click.coords[1] >= object_roi[3]:
                return False
    return True
