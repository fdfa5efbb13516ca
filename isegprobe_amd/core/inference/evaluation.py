"""NoC evaluation loops (reference core/inference/evaluation.py:22-88)."""
from time import time
from typing import Callable, List, Tuple

import numpy as np
import torch

from . import utils
from .clicker import Click, Clicker, DeviceClicker
from .predictors import BasePredictor


def evaluate_dataset(dataset, predictor: BasePredictor, shard: Tuple[int, int] = None, **kwargs) -> Tuple[List[np.ndarray], float]:
    """Per-object IoU arrays in dataset order, and the wall time (reference evaluation.py:22-40).

    ``shard=(rank, world)``: objects of different images are independent, so rank r runs images r, r + world, ... on its own
    GPU -- no data-path collective -- and the per-object arrays are gathered afterwards (``gather_sharded_ious``) into the
    order the single-process loop produces; the time is the slowest rank's.  The reference evaluates on one GPU
    (inference/utils.py:270-274); the arrays, and so every NoC / IoU figure derived from them, are the same either way."""
    rank, world = shard if shard is not None else (0, 1)
    mine = []  # (image index, [IoU array per object])
    start_time = time()
    for index in range(rank, len(dataset), world):
        sample = dataset.get_sample(index)
        per_object = []
        for object_id in sample.objects_ids:
            _, sample_ious, _ = evaluate_sample(sample.image, sample.gt_mask(object_id), predictor,
                                                sample_id=index, **kwargs)
            per_object.append(sample_ious)
        mine.append((index, per_object))
    elapsed = time() - start_time
    if world > 1:
        return gather_sharded_ious(mine, elapsed, len(dataset))
    return [a for _, per_object in mine for a in per_object], elapsed


def gather_sharded_ious(mine, elapsed, n_images):
    """All ranks' (image index, per-object IoU arrays) -> the single-process list (image order, objects in ``objects_ids`` order)
    on every rank, with the slowest rank's time.  A few KB of host objects: ``all_gather_object`` on whatever process group the
    caller initialised."""
    from torch import distributed as dist
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (mine, elapsed))
    by_index = {}
    for part, _ in parts:
        for index, per_object in part:
            if index in by_index:
                raise RuntimeError(f"image {index} was evaluated by two ranks")
            by_index[index] = per_object
    if sorted(by_index) != list(range(n_images)):
        raise RuntimeError(f"sharded evaluation covered {len(by_index)} of {n_images} images")
    return [a for index in range(n_images) for a in by_index[index]], max(t for _, t in parts)


class _HostUser:
    """The reference's arrangement (evaluation.py:43-88): probability map, masks, robot user and IoU on the host."""

    def __init__(self, gt_mask, predictor, pred_thr):
        self.gt_mask, self.predictor, self.thr = gt_mask, predictor, pred_thr
        self.clicker = Clicker(gt_mask=gt_mask)
        self.clicker.make_next_click(np.zeros_like(gt_mask))  # the first click answers an empty prediction
        self.probs = None

    def respond(self):
        """Predict with the clicks so far, score the prediction, let the robot place its next click -> IoU."""
        self.probs = self.predictor.get_prediction(self.clicker)
        mask = self.probs > self.thr
        self._pending = mask
        return utils.get_iou(self.gt_mask, mask)

    def next_click(self):
        self.clicker.make_next_click(self._pending)

    def probs_numpy(self):
        return self.probs


class _DeviceUser:
    """The same interaction with everything but the click bookkeeping on the GPU: one HIP call per click thresholds the
    prediction, counts the IoU and finds the robot's answer; 32 bytes reach the host."""

    def __init__(self, gt_mask, predictor, pred_thr):
        self.predictor, self.thr = predictor, pred_thr
        self.clicker = DeviceClicker(gt_mask=gt_mask, device=predictor.device)
        H, W = gt_mask.shape[:2]
        _, first = self.clicker.evaluate_prediction(torch.zeros(H, W, device=self.clicker.device), pred_thr)
        self.clicker.add_click(first)
        self.probs = None

    def respond(self):
        self.probs = self.predictor.get_prediction_device(self.clicker)
        iou, self._pending = self.clicker.evaluate_prediction(self.probs, self.thr)
        return iou

    def next_click(self):
        self.clicker.add_click(self._pending)

    def probs_numpy(self):
        return self.probs.cpu().numpy()


def evaluate_sample(image: np.ndarray, gt_mask: np.ndarray, predictor: BasePredictor, max_iou_thr: float,
                    pred_thr: float = 0.49, min_clicks: int = 1, max_clicks: int = 20, sample_id: int = None,
                    callback: Callable = None, feats_callback: Callable = None, device_clicker: bool = None
                    ) -> Tuple[List[Click], np.ndarray, np.ndarray]:
    """One object of the NoC protocol (reference evaluation.py:43-88): the robot user clicks into the largest error region of
    the current prediction until the IoU reaches ``max_iou_thr`` (not before ``min_clicks``) or ``max_clicks`` are spent.
    Returns (clicks, IoU after each click, last probability map).  ``device_clicker`` (default: on when the predictor
    runs on a GPU) keeps the probability map, the masks, the robot user and the IoU on the device -- same clicks and IoUs as
    the host arrangement (tests/test_inference_gpu.py).  ``callback(image, gt, probs, sample_id, k, clicks)`` sees every
    prediction; ``feats_callback`` the low- / high-res features before it (it runs first: the prediction moves predictor
    state)."""
    from copy import deepcopy
    if device_clicker is None:
        device_clicker = torch.device(predictor.device).type == "cuda"
    ious = []
    with torch.no_grad():
        predictor.set_input_image(image)
        user = (_DeviceUser if device_clicker else _HostUser)(gt_mask, predictor, pred_thr)
        for k in range(max_clicks):
            if k:
                user.next_click()
            clicks = user.clicker.clicks_list
            if feats_callback is not None:
                _, feats = predictor.get_lowres_highres_feats(deepcopy(user.clicker))
                feats_callback(image, feats, sample_id, k, clicks)
            ious.append(user.respond())
            if callback is not None:
                callback(image, gt_mask, user.probs_numpy(), sample_id, k, clicks)
            if ious[-1] >= max_iou_thr and k + 1 >= min_clicks:
                break
    return user.clicker.clicks_list, np.array(ious, dtype=np.float32), user.probs_numpy()
