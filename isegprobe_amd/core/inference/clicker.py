"""Robot user for NoC evaluation (reference core/inference/clicker.py:11-136).

The next click is the interior-most point (exact Euclidean distance transform on the
1-pixel-padded mask) of the larger of the false-negative / false-positive regions, first
maximum in row-major order.  The reference uses cv2.distanceTransform(DIST_L2, maskSize 0);
OpenCV is not a dependency here, scipy's exact EDT computes the same transform.  ``Clicker`` runs
on the host, as in the reference; ``DeviceClicker`` keeps the masks in HBM and gets the next click
and the IoU of a prediction from one HIP call, syncing 32 bytes per click
instead of two full-resolution device->host copies."""
from copy import deepcopy
from typing import List, Tuple

import numpy as np
import torch
from scipy.ndimage import distance_transform_edt


class Click:
    def __init__(self, is_positive: bool, coords: Tuple[int, int], indx: int = None) -> None:
        self.is_positive = is_positive
        self.coords = coords
        self.indx = indx

    @property
    def coords_and_indx(self):
        return (*self.coords, self.indx)

    def copy(self, **kwargs) -> "Click":
        c = deepcopy(self)
        for k, v in kwargs.items():
            setattr(c, k, v)
        return c


class Clicker(object):
    def __init__(self, gt_mask: np.ndarray = None, init_clicks: List[Click] = None, ignore_label: int = -1,
                 click_indx_offset: int = 0) -> None:
        self.click_indx_offset = click_indx_offset
        if gt_mask is not None:
            self.gt_mask = gt_mask == 1
            self.not_ignore_mask = gt_mask != ignore_label
        else:
            self.gt_mask = None
        self.reset_clicks()
        for click in init_clicks or ():
            self.add_click(click)

    def make_next_click(self, pred_mask: np.ndarray) -> None:
        assert self.gt_mask is not None
        self.add_click(self._get_next_click(pred_mask))

    def get_clicks(self, clicks_limit: int = None) -> List[Click]:
        return self.clicks_list[:clicks_limit]

    def _get_next_click(self, pred_mask: np.ndarray, padding: bool = True) -> Click:
        pred = pred_mask.astype(bool)
        fn_mask = self.gt_mask & ~pred & self.not_ignore_mask
        fp_mask = ~self.gt_mask & pred & self.not_ignore_mask

        def interior_distance(mask):
            if padding:
                mask = np.pad(mask, ((1, 1), (1, 1)), "constant")
            dt = distance_transform_edt(mask).astype(np.float32)  # cv2.distanceTransform returns float32
            return dt[1:-1, 1:-1] if padding else dt

        fn_dt = interior_distance(fn_mask) * self.not_clicked_map
        fp_dt = interior_distance(fp_mask) * self.not_clicked_map
        fn_max, fp_max = np.max(fn_dt), np.max(fp_dt)
        is_positive = fn_max > fp_max
        ys, xs = np.where((fn_dt == fn_max) if is_positive else (fp_dt == fp_max))
        return Click(is_positive=is_positive, coords=(ys[0], xs[0]))

    def add_click(self, click: Click) -> None:
        coords = click.coords
        click.indx = self.click_indx_offset + self.num_pos_clicks + self.num_neg_clicks
        if click.is_positive:
            self.num_pos_clicks += 1
        else:
            self.num_neg_clicks += 1
        self.clicks_list.append(click)
        if self.gt_mask is not None:
            self.not_clicked_map[coords[0], coords[1]] = False

    def _remove_last_click(self) -> None:
        click = self.clicks_list.pop()
        if click.is_positive:
            self.num_pos_clicks -= 1
        else:
            self.num_neg_clicks -= 1
        if self.gt_mask is not None:
            self.not_clicked_map[click.coords[0], click.coords[1]] = True

    def reset_clicks(self) -> None:
        if self.gt_mask is not None:
            self.not_clicked_map = np.ones_like(self.gt_mask, dtype=bool)
        self.num_pos_clicks = 0
        self.num_neg_clicks = 0
        self.clicks_list = []

    def get_state(self) -> List[Click]:
        return deepcopy(self.clicks_list)

    def set_state(self, state: List[Click]) -> None:
        self.reset_clicks()
        for click in state:
            self.add_click(click)

    def __len__(self) -> int:
        return len(self.clicks_list)


class DeviceClicker(Clicker):
    """Clicker whose masks live on the GPU.  ``evaluate_prediction(probs, thr)`` thresholds a device
    probability map, computes IoU (utils.get_iou) and the robot's next click (``_get_next_click``) in
    one HIP call, and returns ``(iou, Click)`` after a 32-byte copy; ``make_next_click`` accepts a
    device (bool/uint8) or host mask.  Click bookkeeping (lists, counters, states) is the host
    class's; the not-clicked map is mirrored on the device."""

    def __init__(self, gt_mask: np.ndarray = None, init_clicks: List[Click] = None, ignore_label: int = -1,
                 click_indx_offset: int = 0, device="cuda") -> None:
        self.device = torch.device(device)
        if gt_mask is None:
            raise ValueError("DeviceClicker needs a ground-truth mask")
        gt = np.asarray(gt_mask)
        self._gt_dev = torch.from_numpy(np.ascontiguousarray(gt == 1).astype(np.uint8)).to(self.device)
        self._ni_dev = torch.from_numpy(np.ascontiguousarray(gt != ignore_label).astype(np.uint8)).to(self.device)
        self._nc_dev = torch.ones_like(self._gt_dev)
        self._ws = None
        super().__init__(gt_mask, init_clicks, ignore_label, click_indx_offset)

    # ---- device state mirrors
    def add_click(self, click: Click) -> None:
        super().add_click(click)
        self._nc_dev[int(click.coords[0]), int(click.coords[1])] = 0

    def _remove_last_click(self) -> None:
        click = self.clicks_list[-1]
        super()._remove_last_click()
        self._nc_dev[int(click.coords[0]), int(click.coords[1])] = 1

    def reset_clicks(self) -> None:
        super().reset_clicks()
        if hasattr(self, "_nc_dev"):
            self._nc_dev.fill_(1)

    # ---- the fused device step
    def _device_mask(self, pred_mask):
        if isinstance(pred_mask, np.ndarray):
            pred_mask = torch.from_numpy(np.ascontiguousarray(pred_mask))
        return (pred_mask.to(self.device) != 0).to(torch.uint8).contiguous()

    def _run(self, pred_u8):
        from ... import hip_ops
        rec, self._ws = hip_ops.robot_click(pred_u8, self._gt_dev, self._ni_dev, self._nc_dev, self._ws)
        is_pos, y, x, _, _, inter, union, _ = rec.tolist()  # the only device->host copy of the click step
        iou = inter / union if union > 0 else float("nan")  # numpy's 0/0 in utils.get_iou
        return iou, Click(is_positive=bool(is_pos), coords=(y, x))

    def evaluate_prediction(self, probs: torch.Tensor, pred_thr: float):
        """probs: device f32 [H,W] probability map -> (IoU of probs > thr, the click the robot would make next)."""
        from ... import hip_ops
        return self._run(hip_ops.threshold_u8(probs, pred_thr))

    def _get_next_click(self, pred_mask, padding: bool = True) -> Click:
        if not padding:
            raise NotImplementedError("the device clicker implements the padded transform the evaluation uses")
        return self._run(self._device_mask(pred_mask))[1]
