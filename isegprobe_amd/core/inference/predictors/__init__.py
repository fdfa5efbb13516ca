"""Predictor factory (reference core/inference/predictors/__init__.py:15-119).  Only the NoBRS
predictor is on the probed path (every experiment of the reference used NoBRS,
configs/eval_cfg.yaml:13-14); the BRS modes are out of scope."""
from typing import Dict

from ..transforms import ZoomIn
from .base_predictor import BasePredictor


def get_predictor(net, brs_mode: str, device, prob_thresh: float = 0.49, with_flip: bool = True,
                  zoom_in_params: Dict = dict(), predictor_params: Dict = None, brs_opt_func_params: Dict = None,
                  lbfgs_params: Dict = None) -> BasePredictor:
    predictor_params_ = {"optimize_after_n_clicks": 1}
    zoom_in = ZoomIn(**zoom_in_params) if zoom_in_params is not None else None
    if brs_mode != "NoBRS":
        raise NotImplementedError(f"brs_mode={brs_mode}: BRS refinement is outside the dense-feature path")
    if predictor_params is not None:
        predictor_params_.update(predictor_params)
    return BasePredictor(net, device, zoom_in=zoom_in, with_flip=with_flip, **predictor_params_)


__all__ = ["BasePredictor", "get_predictor"]
