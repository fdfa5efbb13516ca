"""Bounding-box helpers used by the zoom-in transform (reference core/utils/misc.py:71-119)."""
from typing import Tuple

import numpy as np


def get_bbox_from_mask(mask: np.ndarray) -> Tuple[int, int, int, int]:
    rows = np.where(np.any(mask, axis=1))[0]
    cols = np.where(np.any(mask, axis=0))[0]
    return rows[0], rows[-1], cols[0], cols[-1]


def expand_bbox(bbox: Tuple, expand_ratio: float, min_crop_size: int = None) -> Tuple:
    rmin, rmax, cmin, cmax = bbox
    rcenter, ccenter = 0.5 * (rmin + rmax), 0.5 * (cmin + cmax)
    height = expand_ratio * (rmax - rmin + 1)
    width = expand_ratio * (cmax - cmin + 1)
    if min_crop_size is not None:
        height, width = max(height, min_crop_size), max(width, min_crop_size)
    return (int(round(rcenter - 0.5 * height)), int(round(rcenter + 0.5 * height)),
            int(round(ccenter - 0.5 * width)), int(round(ccenter + 0.5 * width)))


def clamp_bbox(bbox: Tuple, rmin: float, rmax: float, cmin: float, cmax: float) -> Tuple:
    return max(rmin, bbox[0]), min(rmax, bbox[1]), max(cmin, bbox[2]), min(cmax, bbox[3])


def get_segments_iou(s1: Tuple, s2: Tuple) -> float:
    a, b = s1
    c, d = s2
    intersection = max(0, min(b, d) - max(a, c) + 1)
    union = max(1e-6, max(b, d) - min(a, c) + 1)
    return intersection / union


def get_bbox_iou(b1, b2):
    return get_segments_iou(b1[:2], b2[:2]) * get_segments_iou(b1[2:4], b2[2:4])
