"""CPU: host logic of the inference mirror (clicker, IoU, NoC, bbox helpers) against the
known answers generated from the reference (tests/golden/inference.npz)."""
import numpy as np

from isegprobe_amd.core.inference.clicker import Click, Clicker
from isegprobe_amd.core.inference.utils import compute_noc_metric, get_iou


def test_clicker_sequence_and_iou(golden):
    g = golden("inference")
    gt = g["gt"]
    pred = np.zeros_like(gt, dtype=bool)
    ck = Clicker(gt_mask=gt)
    seq = []
    for _ in range(4):
        ck.make_next_click(pred)
        c = ck.get_clicks()[-1]
        seq.append((c.coords[0], c.coords[1], int(c.is_positive)))
        pred = pred.copy()
        pred[max(0, c.coords[0] - 12):c.coords[0] + 12, max(0, c.coords[1] - 30):c.coords[1] + 9] = c.is_positive
    assert np.array_equal(np.array(seq, dtype=np.int64), g["clicker_seq"])
    assert np.array_equal(pred, g["clicker_final_pred"])
    assert get_iou(gt, pred) == float(g["iou_value"])


def test_noc_metric_known_answer(golden):
    g = golden("inference")
    all_ious = [np.array([0.3, 0.85, 0.91, 0.95]), np.array([0.5, 0.6]), np.array([0.92])]
    noc, std, over = compute_noc_metric(all_ious, [0.8, 0.85, 0.9], max_clicks=20)
    assert np.allclose(noc, g["noc"]) and np.allclose(std, g["noc_std"]) and np.array_equal(over, g["noc_over"])
    # by inspection: thr 0.9 -> clicks 3, 20 (never), 1
    assert noc[2] == (3 + 20 + 1) / 3 and over[2] == 1


def test_points_nd_layout():
    import torch
    from isegprobe_amd.core.inference.predictors.base_predictor import BasePredictor
    p = BasePredictor(model=None, device=torch.device("cpu"))
    clicks = [Click(True, (3, 4), 0), Click(False, (7, 8), 1), Click(True, (1, 2), 2)]
    pts = p.get_points_nd([clicks])
    assert pts.shape == (1, 4, 3)
    assert pts[0].tolist() == [[3, 4, 0], [1, 2, 2], [7, 8, 1], [-1, -1, -1]]


def test_bbox_helpers():
    from isegprobe_amd.core.utils.misc import clamp_bbox, expand_bbox, get_bbox_from_mask, get_bbox_iou
    m = np.zeros((20, 30), bool)
    m[5:9, 10:21] = True
    assert get_bbox_from_mask(m) == (5, 8, 10, 20)
    assert expand_bbox((5, 8, 10, 20), 1.4, 6) == (4, 10, 7, 23)   # height max(5.6, 6) = 6, width 15.4
    assert clamp_bbox((-3, 25, 2, 40), 0, 19, 0, 29) == (0, 19, 2, 29)
    assert abs(get_bbox_iou((0, 9, 0, 9), (5, 14, 0, 9)) - (5 / 15)) < 1e-9


def test_jbu_resize_fusion_condition():
    """Host logic of the fused last-JBU-stage + resize: only the 8:7 geometry (FeatUp's x16 map vs a patch-14 image)
    qualifies, and within it src = dst * (GH-1)/(OH-1) never leaves the 8-row group of its 7 output rows -- the property
    the no-halo blend in jbu_kernels_kernel<BLEND> / jbu_apply_resized_kernel relies on (checked here in the float32
    arithmetic the kernels use)."""
    import numpy as np
    from isegprobe_amd.core.model.upsamplers.JBUFeatUp import JBULearnedRange
    ok = JBULearnedRange.resize_fusable
    assert ok(512, 512, 448, 448) and ok(256, 384, 224, 336) and ok(64, 128, 56, 112)
    assert not ok(512, 512, 447, 448) and not ok(512, 512, 512, 512) and not ok(500, 512, 448, 448) and not ok(36, 36, 32, 31)
    for m in (1, 2, 5, 9, 16, 32, 64, 100):
        GH, OH = 8 * m, 7 * m
        s = np.float32(GH - 1) / np.float32(OH - 1) if OH > 1 else np.float32(0)
        Y = np.arange(OH, dtype=np.float32)
        y0 = (s * Y).astype(np.int32)
        y1 = np.minimum(y0 + 1, GH - 1)
        grp = np.arange(OH) // 7
        assert (y0 // 8 == grp).all() and (y1 // 8 == grp).all(), m
        # and the 16-column group of a 14-pixel strip
        assert (y0 // 16 == np.arange(OH) // 14).all() and (y1 // 16 == np.arange(OH) // 14).all(), m


def test_eval_mode_zoom_in_params():
    """core/inference/utils.py:301-316 (eval_ritm=False): fixed<H>, fixed<H>,<W>, cvpr (448 / DAVIS 672)."""
    import pytest
    from isegprobe_amd.core.inference.utils import get_zoom_in_params
    assert get_zoom_in_params("fixed224") == {"skip_clicks": -1, "target_size": (224, 224)}
    assert get_zoom_in_params("fixed400,600", "SBD") == {"skip_clicks": -1, "target_size": (400, 600)}
    assert get_zoom_in_params("cvpr", "GrabCut") == {"skip_clicks": -1, "target_size": (448, 448)}
    assert get_zoom_in_params("cvpr", "DAVIS") == {"skip_clicks": -1, "target_size": (672, 672)}
    with pytest.raises(NotImplementedError):
        get_zoom_in_params("original")

