"""Synthetic training clicks (reference core/data/points_sampler.py:35-380, the configuration the SBD scripts use:
models/defaults.py:74-79 -- MultiPointSampler(num_max_points, prob_gamma=0.8, merge_objects_prob=0.15,
max_num_merged_objects=2); no object hierarchy, no soft targets, no "first click at the centre").

Per sample: pick one object (or, with probability merge_objects_prob, merge up to max_num_merged_objects of them) as the
target mask; draw 1 + Geometric-like(prob_gamma) positive clicks uniformly from the (usually eroded) object and
0..max Geometric-like negative clicks from a mixture of three regions -- anywhere outside the object ("bg", 0.1), other
objects ("other", 0.4), a band around the object ("border", 0.5).  Output: ``[2 * max_num_points, 3]`` rows
(row, col, 100), padded with (-1, -1, -1): positives first, as DistMaps expects (core/model/ops.py:35-77).

OpenCV is absent from the image: the 3 x 3 erosion / dilation of cv2.erode / cv2.dilate (default border handling: the
outside counts as set for erosion, as clear for dilation) are scipy.ndimage's binary morphology with the same structuring
element and iteration count."""
import math
import random
from functools import lru_cache
from typing import List

import numpy as np
from scipy import ndimage

_K3 = np.ones((3, 3), bool)


@lru_cache(maxsize=None)
def generate_probs(max_num_points: int, gamma: float) -> np.ndarray:
    """points_sampler.py:341-352: p(i) ~ gamma^i, i = 0 .. max_num_points - 1."""
    p = gamma ** np.arange(max_num_points, dtype=np.float64)
    return p / p.sum()


class MultiPointSampler:
    neg_strategies = ("bg", "other", "border")

    def __init__(self, max_num_points: int, prob_gamma: float = 0.7, expand_ratio: float = 0.1, positive_erode_prob: float = 0.9,
                 positive_erode_iters: int = 3, negative_bg_prob: float = 0.1, negative_other_prob: float = 0.4,
                 negative_border_prob: float = 0.5, merge_objects_prob: float = 0.0, max_num_merged_objects: int = 2) -> None:
        self.max_num_points = max_num_points
        self.expand_ratio = expand_ratio
        self.positive_erode_prob, self.positive_erode_iters = positive_erode_prob, positive_erode_iters
        self.merge_objects_prob = merge_objects_prob
        self.max_num_merged_objects = max_num_points if max_num_merged_objects == -1 else max_num_merged_objects
        self.neg_strategies_prob = [negative_bg_prob, negative_other_prob, negative_border_prob]
        assert math.isclose(sum(self.neg_strategies_prob), 1.0)
        self._pos_probs = generate_probs(max_num_points, prob_gamma)
        self._neg_probs = generate_probs(max_num_points + 1, prob_gamma)
        self._selected_mask = self._pos_masks = self._neg_regions = None

    # ------------------------------------------------------------------ object choice
    @property
    def selected_mask(self) -> np.ndarray:
        """[1, H, W] float32 target of the sample (points_sampler.py:24-32)."""
        assert self._selected_mask is not None
        return self._selected_mask

    def sample_object(self, sample) -> None:
        """points_sampler.py:84-117.  ``sample``: a DSample-like with __len__, objects_ids, get_object_mask(i) and
        get_background_mask()."""
        if len(sample) == 0:
            bg = sample.get_background_mask()
            self._selected_mask = np.zeros((1,) + bg.shape, np.float32)
            self._pos_masks = [[]]
            self._neg_regions = [bg, bg, bg]
            return
        ids = list(sample.objects_ids)
        if len(ids) > 1 and random.random() < self.merge_objects_prob:
            k = np.random.randint(2, min(len(ids), self.max_num_merged_objects) + 1)
            chosen = random.sample(ids, k)
        else:
            chosen = [random.choice(ids)]
        masks = [sample.get_object_mask(i) > 0 for i in chosen]
        gt = np.logical_or.reduce(masks)
        self._selected_mask = gt[None].astype(np.float32)
        self._pos_masks = [self._positive_erode(m) for m in masks]
        not_gt = ~gt
        other = not_gt if len(sample) <= len(masks) else (~sample.get_background_mask()) & not_gt
        self._neg_regions = [not_gt, other, self._border_mask(gt)]

    def _positive_erode(self, mask: np.ndarray) -> np.ndarray:
        """points_sampler.py:318-331."""
        if random.random() > self.positive_erode_prob:
            return mask
        eroded = ndimage.binary_erosion(mask, _K3, iterations=self.positive_erode_iters, border_value=1)
        return eroded if eroded.sum() > 10 else mask

    def _border_mask(self, mask: np.ndarray) -> np.ndarray:
        """points_sampler.py:333-338: a band of width ceil(expand_ratio * sqrt(area)) around the object."""
        r = int(np.ceil(self.expand_ratio * np.sqrt(mask.sum())))
        grown = ndimage.binary_dilation(mask, _K3, iterations=r, border_value=0) if r > 0 else mask.copy()
        grown[mask] = False
        return grown

    # ------------------------------------------------------------------ clicks
    def sample_points(self) -> List:
        """points_sampler.py:206-222: max_num_points positive rows, then max_num_points negative rows."""
        pos = self._group([(m, False) for m in self._pos_masks])
        neg = self._group([(list(zip(self._neg_regions, self.neg_strategies_prob)), True)])
        return pos + neg

    def _group(self, masks) -> List:
        """points_sampler.py:224-270 for one polarity: one mask -> its own draw; several (merged objects) -> the first
        click of each, then a draw from their union."""
        masks = masks[: self.max_num_points]
        per = [self._draw(m, neg) for m, neg in masks]
        per = [p for p in per if p]
        points = []
        if len(per) == 1:
            points = per[0]
        elif len(per) > 1:
            points = [p[0] for p in per]
            union = [(m, 1.0 / len(masks)) for m, _ in masks]
            extra = self._draw(union, True)
            room = self.max_num_points - len(points)
            points.extend(extra if len(extra) <= room else random.sample(extra, room))
        points = points[: self.max_num_points]
        return points + [(-1, -1, -1)] * (self.max_num_points - len(points))

    def _draw(self, mask, is_negative: bool) -> List:
        """points_sampler.py:272-316: the number of clicks, then uniform draws (from a mixture of regions when ``mask`` is
        a list of (region, probability))."""
        if isinstance(mask, list) and not mask:
            return []
        if is_negative:
            n = np.random.choice(np.arange(self.max_num_points + 1), p=self._neg_probs)
        else:
            n = 1 + np.random.choice(np.arange(self.max_num_points), p=self._pos_probs)
        mixture = isinstance(mask, list)
        if mixture:
            regions, probs = [np.argwhere(m) for m, _ in mask], [p for _, p in mask]
            assert math.isclose(sum(probs), 1.0)
        else:
            regions, probs = [np.argwhere(mask)], None
        points = []
        for _ in range(n):
            idx = regions[np.random.choice(len(regions), p=probs)] if mixture else regions[0]
            if len(idx):
                points.append(idx[np.random.randint(0, len(idx))].tolist() + [100])
        return points
