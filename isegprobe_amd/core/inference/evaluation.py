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


def evaluate_sample(image: np.ndarray, gt_mask: np.ndarray, predictor: BasePredictor, max_iou_thr: float,
                    pred_thr: float = 0.49, min_clicks: int = 1, max_clicks: int = 20, sample_id: int = None,
                    callback: Callable = None, feats_callback: Callable = None, device_clicker: bool = None
                    ) -> Tuple[List[Click], np.ndarray, np.ndarray]:
    """``device_clicker`` (default: on when the predictor runs on a GPU) keeps the probability map, the
    masks, the robot user and the IoU on the device -- same clicks and IoUs as the host path
    (tests/test_inference_gpu.py), 32 bytes copied per click."""
    from copy import deepcopy
    if device_clicker is None:
        device_clicker = torch.device(predictor.device).type == "cuda"
    if device_clicker:
        return _evaluate_sample_device(image, gt_mask, predictor, max_iou_thr, pred_thr, min_clicks, max_clicks,
                                       sample_id, callback, feats_callback)
    clicker = Clicker(gt_mask=gt_mask)
    pred_mask = np.zeros_like(gt_mask)
    ious_list = []
    with torch.no_grad():
        predictor.set_input_image(image)
        for click_indx in range(max_clicks):
            clicker.make_next_click(pred_mask)
            if feats_callback is not None:  # before get_prediction: it changes the predictor state
                _, feats = predictor.get_lowres_highres_feats(deepcopy(clicker))
                feats_callback(image, feats, sample_id, click_indx, clicker.clicks_list)
            pred_probs = predictor.get_prediction(clicker)
            pred_mask = pred_probs > pred_thr
            if callback is not None:
                callback(image, gt_mask, pred_probs, sample_id, click_indx, clicker.clicks_list)
            iou = utils.get_iou(gt_mask, pred_mask)
            ious_list.append(iou)
            if iou >= max_iou_thr and click_indx + 1 >= min_clicks:
                break
    return clicker.clicks_list, np.array(ious_list, dtype=np.float32), pred_probs


def _evaluate_sample_device(image, gt_mask, predictor, max_iou_thr, pred_thr, min_clicks, max_clicks, sample_id, callback,
                            feats_callback):
    from copy import deepcopy
    clicker = DeviceClicker(gt_mask=gt_mask, device=predictor.device)
    ious_list = []
    with torch.no_grad():
        predictor.set_input_image(image)
        H, W = gt_mask.shape[:2]
        # first click: the robot looks at an empty prediction (evaluation.py:61,66)
        _, click = clicker.evaluate_prediction(torch.zeros(H, W, device=clicker.device), pred_thr)
        for click_indx in range(max_clicks):
            clicker.add_click(click)
            if feats_callback is not None:
                _, feats = predictor.get_lowres_highres_feats(deepcopy(clicker))
                feats_callback(image, feats, sample_id, click_indx, clicker.clicks_list)
            probs = predictor.get_prediction_device(clicker)
            iou, click = clicker.evaluate_prediction(probs, pred_thr)
            if callback is not None:
                callback(image, gt_mask, probs.cpu().numpy(), sample_id, click_indx, clicker.clicks_list)
            ious_list.append(iou)
            if iou >= max_iou_thr and click_indx + 1 >= min_clicks:
                break
    return clicker.clicks_list, np.array(ious_list, dtype=np.float32), probs.cpu().numpy()
