"""CPU: the Hydra-style override parser behind train.py / evaluate.py (reference README.md:85-103 invocations)."""
import pytest

from isegprobe_amd.core.utils.overrides import EVAL_DEFAULTS, TRAIN_DEFAULTS, apply_overrides, split_overrides


def test_reference_readme_invocations():
    ov, rest = split_overrides(["+exp.name=my_name", "+exp.model_path=/path/to/my/model", "--steps", "3", "dataloader.batch_size=32",
                                "training_params.crop_size=[448,448]"])
    assert rest == ["--steps", "3"]
    cfg = apply_overrides(TRAIN_DEFAULTS, ov)
    assert cfg["exp"] == {"name": "my_name", "model_path": "/path/to/my/model"}
    assert cfg["dataloader"]["batch_size"] == 32 and cfg["training_params"]["crop_size"] == [448, 448]
    assert TRAIN_DEFAULTS["dataloader"]["batch_size"] == 8  # defaults untouched
    ov, rest = split_overrides(["+checkpoint=/path/to/checkpoints", "+datasets=GrabCut,Berkeley,SBD,DAVIS", "thresh=0.49",
                                "print_ious=false", "n_clicks=10"])
    cfg = apply_overrides(EVAL_DEFAULTS, ov)
    assert cfg["checkpoint"] == "/path/to/checkpoints" and cfg["datasets"] == "GrabCut,Berkeley,SBD,DAVIS"
    assert cfg["thresh"] == 0.49 and cfg["print_ious"] is False and cfg["n_clicks"] == 10 and not rest


def test_unknown_key_needs_plus():
    ov, _ = split_overrides(["no_such_key=1"])
    with pytest.raises(SystemExit):
        apply_overrides(EVAL_DEFAULTS, ov)
    ov, _ = split_overrides(["+no_such_key=1", "++a.b.c=x"])
    cfg = apply_overrides(EVAL_DEFAULTS, ov)
    assert cfg["no_such_key"] == 1 and cfg["a"]["b"]["c"] == "x"


def test_eval_config_keys_of_the_click_budget():
    """configs/eval_cfg.yaml keys evaluate.py honours beyond the README's: clicks_limit (-1 = n_clicks -> predictor
    net_clicks_limit, inference/utils.py:286-289), min_n_clicks and iou_analysis (both decide whether every click runs,
    inference/utils.py:254-257)."""
    ov, rest = split_overrides(["clicks_limit=-1", "min_n_clicks=2", "iou_analysis=true", "n_clicks=8"])
    cfg = apply_overrides(EVAL_DEFAULTS, ov)
    assert cfg["clicks_limit"] == -1 and cfg["min_n_clicks"] == 2 and cfg["iou_analysis"] is True and cfg["n_clicks"] == 8 and not rest
    assert EVAL_DEFAULTS["clicks_limit"] is None and EVAL_DEFAULTS["min_n_clicks"] == 1 and EVAL_DEFAULTS["iou_analysis"] is False
