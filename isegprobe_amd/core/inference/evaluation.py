"""NoC evaluation loops (reference core/inference/evaluation.py:22-88)."""
from time import time
from typing import Callable, List, Tuple

import numpy as np
import torch

from . import utils
from .clicker import Click, Clicker, DeviceClicker
from .predictors import BasePredictor


def evaluate_dataset(dataset, predictor: BasePredictor, **kwargs) -> Tuple[List[np.ndarray], float]:
    all_ious = []
    start_time = time()
    for index in range(len(dataset)):
        sample = dataset.get_sample(index)
        for object_id in sample.objects_ids:
            _, sample_ious, _ = evaluate_sample(sample.image, sample.gt_mask(object_id), predictor,
                                                sample_id=index, **kwargs)
            all_ious.append(sample_ious)
    return all_ious, time() - start_time


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
