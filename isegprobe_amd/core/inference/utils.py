"""IoU / NoC metrics (reference core/inference/utils.py:107-146)."""
from typing import List, Tuple

import numpy as np


def get_iou(gt_mask: np.ndarray, pred_mask: np.ndarray, ignore_label: int = -1) -> float:
    keep = gt_mask != ignore_label
    obj = gt_mask == 1
    intersection = np.logical_and(np.logical_and(pred_mask, obj), keep).sum()
    union = np.logical_and(np.logical_or(pred_mask, obj), keep).sum()
    return intersection / union


def compute_noc_metric(all_ious: List[np.ndarray], iou_thrs: List[float], max_clicks: int = 20
                       ) -> Tuple[List[float], List[float], List[int]]:
    def noc(iou_arr, thr):
        hit = iou_arr >= thr
        return np.argmax(hit) + 1 if np.any(hit) else max_clicks

    noc_list, noc_std, over_max = [], [], []
    for thr in iou_thrs:
        scores = np.array([noc(a, thr) for a in all_ious], dtype=np.int_)
        noc_list.append(scores.mean())
        noc_std.append(scores.std())
        over_max.append((scores == max_clicks).sum())
    return noc_list, noc_std, over_max
