"""Checkpoint writer and bounding-box helpers (reference core/utils/misc.py:37-119)."""
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


def save_checkpoint(net, checkpoints_path, epoch=None, prefix="", verbose=True, multi_gpu=False):
    """core/utils/misc.py:37-68: ``{"state_dict": net.get_state_dict_to_save(), "config": net._config}`` as
    ``[<prefix>_]last_checkpoint.pth`` (epoch None) or ``[<prefix>_]<epoch:03d>.pth``.  The file loads in the reference
    and the reference's checkpoints load here (class path ``core.model...``, see isegprobe_amd.install_as_core)."""
    from pathlib import Path
    import torch
    checkpoints_path = Path(checkpoints_path)
    name = "last_checkpoint.pth" if epoch is None else f"{epoch:03d}.pth"
    if prefix:
        name = f"{prefix}_{name}"
    checkpoints_path.mkdir(parents=True, exist_ok=True)
    net = net.module if multi_gpu else net
    state_dict = net.get_state_dict_to_save() if hasattr(net, "get_state_dict_to_save") else net.state_dict()
    config = dict(net._config)
    # reference-compatible class path (serialization.py:65): the reference resolves "core.model.iseg_probe_model..."
    config["class"] = config["class"].replace("isegprobe_amd.core.", "core.")
    torch.save({"state_dict": state_dict, "config": config}, str(checkpoints_path / name))
    if verbose:
        print(f"Save checkpoint to {checkpoints_path / name}")
    return checkpoints_path / name
