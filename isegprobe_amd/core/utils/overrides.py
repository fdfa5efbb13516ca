"""Hydra-style command-line overrides without hydra (absent from the image).

The reference's entry points are Hydra scripts (train.py:13-27, evaluate.py:30-31) driven as
``python train.py +exp.name=my_name +exp.model_path=models/sbd/dinov2/patch-embed_loftup.py`` and
``python evaluate.py +checkpoint=/path/to/ckpt +datasets=GrabCut,Berkeley`` (README.md:85-103) on top of
configs/train_cfg.yaml / eval_cfg.yaml.  ``split_overrides`` separates such tokens (``key=value``, ``+key=value``,
``++key=value``, dotted keys) from ordinary ``--flag`` arguments and ``apply_overrides`` lays them over a nested
default config; values are parsed as YAML scalars / lists, as Hydra does for simple values."""
import copy
import re
from typing import Dict, List, Tuple

import yaml

_TOKEN = re.compile(r"^(\+{0,2})([A-Za-z_][\w.\-]*)=(.*)$", re.S)


def split_overrides(argv: List[str]) -> Tuple[List[Tuple[str, object, bool]], List[str]]:
    """-> ([(dotted key, parsed value, may_add)], remaining argv).  A bare ``key=value`` must name an existing key
    (Hydra refuses it otherwise); ``+key=value`` / ``++key=value`` may add one."""
    found, rest = [], []
    for tok in argv:
        m = None if tok.startswith("-") else _TOKEN.match(tok)
        if m is None:
            rest.append(tok)
            continue
        plus, key, raw = m.groups()
        try:
            val = yaml.safe_load(raw) if raw != "" else ""
        except yaml.YAMLError:
            val = raw
        if isinstance(val, str) and "," in val and not raw.startswith(("'", '"')):
            val = raw  # "GrabCut,Berkeley": the reference splits such strings itself (evaluate.py:52)
        found.append((key, val, bool(plus)))
    return found, rest


def apply_overrides(defaults: Dict, overrides) -> Dict:
    cfg = copy.deepcopy(defaults)
    for key, val, may_add in overrides:
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node or not isinstance(node[p], dict):
                if not may_add:
                    raise SystemExit(f"override '{key}': no such config group '{p}' (prefix the key with + to add it)")
                node[p] = {}
            node = node[p]
        if parts[-1] not in node and not may_add:
            raise SystemExit(f"override '{key}': key not in the config (prefix it with + to add it)")
        node[parts[-1]] = val
    return cfg


# the reference's YAML defaults that matter on this path (configs/train_cfg.yaml, configs/eval_cfg.yaml)
TRAIN_DEFAULTS = {
    "exp": {"name": "exp_name_test", "model_path": "models/sbd/dinov2/patch-embed_bilinear.py"},
    "dataloader": {"workers": 4, "batch_size": 8},
    "training_params": {"epochs": 20, "crop_size": [224, 224], "num_max_points": 24, "do_validation": True,
                        "checkpoint_interval": [[0, 3], [15, 1]], "lr_milestones": [17, 20]},
    "training": {"seed": 0, "ngpus": 1, "gpus": "", "resume_exp": None, "resume_prefix": "latest", "start_epoch": 0,
                 "weights": None, "local_rank": 0, "distributed": False},
    "wandb": {"log_wandb": False, "project": "iSegProbe-Train", "name": "", "dir": ""},
}
EVAL_DEFAULTS = {
    "mode": "NoBRS", "checkpoint": None, "exp_path": "", "datasets": "GrabCut,Berkeley,SBD,DAVIS", "gpus": "0", "cpu": False,
    "target_iou": 0.90, "iou_analysis": False, "n_clicks": 20, "min_n_clicks": 1, "thresh": 0.5, "clicks_limit": None,
    "eval_mode": "fixed224", "eval_ritm": False, "save_ious": False, "print_ious": True, "vis_preds": False,
    "save_feats": False, "save_feats_folder_name": "features", "save_feats_for_n_imgs": 50, "model_name": None, "main_cfg_path": "./configs/main_cfg.yaml", "logs_path": "", "wandb": False,
}
DATASET_PATH_KEYS = {"GrabCut": "GRABCUT_PATH", "Berkeley": "BERKELEY_PATH", "DAVIS": "DAVIS_PATH", "SBD": "SBD_PATH",
                     "SBD_Train": "SBD_PATH", "PascalVOC": "PASCALVOC_PATH", "COCO_MVal": "COCO_MVAL_PATH"}  # inference/utils.py:86-104
