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
