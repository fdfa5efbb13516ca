"""Oracle (test infrastructure only): train-time click simulation, reference core/training/trainer.py:575-618
(get_next_points).  cv2.distanceTransform(mask, DIST_L2, 5) is the C restatement in oracle/chamfer5.c
(PARITY UNPINNED: OpenCV is not in the container).  The reference draws the click uniformly among the inner
pixels with the global numpy RNG; here the draw is explicit (`rand32[b]`, a 32-bit integer per sample:
index = (rand32 * n) >> 32) so that the device path can be compared bit for bit."""
import ctypes
import os

import numpy as np

_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libchamfer5.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not built; run `make -C oracle` or __graft_entry__.build()")
        lib = ctypes.CDLL(path)
        lib.oracle_chamfer5.restype = ctypes.c_int
        lib.oracle_chamfer5.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _LIB = lib
    return _LIB


def chamfer5(mask):
    """mask [h, w] (bool / uint8) -> float32 [h, w], cv2.distanceTransform(mask, cv2.DIST_L2, 5)."""
    m = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty(m.shape, dtype=np.float32)
    if _lib().oracle_chamfer5(m.ctypes.data, m.shape[0], m.shape[1], out.ctypes.data) != 0:
        raise MemoryError("oracle_chamfer5")
    return out


def get_next_points(pred, gt, points, click_indx, rand32, pred_thresh=0.49):
    """pred [B,1,H,W] f32 probabilities, gt [B,1,H,W], points [B,2P,3] (numpy) -> updated copy of points.
    trainer.py:583-618 with the uniform draw made explicit."""
    assert click_indx > 0
    pred = np.asarray(pred)[:, 0]
    gt = np.asarray(gt)[:, 0] > 0.5
    fn_mask = np.pad(np.logical_and(gt, pred < pred_thresh), ((0, 0), (1, 1), (1, 1)), "constant").astype(np.uint8)
    fp_mask = np.pad(np.logical_and(~gt, pred > pred_thresh), ((0, 0), (1, 1), (1, 1)), "constant").astype(np.uint8)
    num_points = points.shape[1] // 2
    points = np.array(points, dtype=np.float32, copy=True)
    for b in range(fn_mask.shape[0]):
        fn_dt = chamfer5(fn_mask[b])[1:-1, 1:-1]
        fp_dt = chamfer5(fp_mask[b])[1:-1, 1:-1]
        fn_max, fp_max = np.max(fn_dt), np.max(fp_dt)
        is_positive = fn_max > fp_max
        dt = fn_dt if is_positive else fp_dt
        indices = np.argwhere(dt > max(fn_max, fp_max) / 2.0)
        if len(indices) > 0:
            r, c = indices[(int(rand32[b]) * len(indices)) >> 32]
            slot = (num_points if is_positive else 2 * num_points) - click_indx
            points[b, slot] = (float(r), float(c), float(click_indx))
    return points
