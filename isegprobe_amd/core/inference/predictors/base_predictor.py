"""Per-click prediction driver (reference core/inference/predictors/base_predictor.py:20-235)."""
from typing import Dict, List, Tuple, Union

import numpy as np
import torch

from .... import hip_ops as ops
from ...model._guidance_cache import guidance_scope, storage_epoch
from ..clicker import Click, Clicker
from ..transforms import AddHorizontalFlip, BaseTransform, LimitLongestSide, SigmoidForPred


class _ClickGraph:
    """One captured HIP graph of ``net(image, points)`` for a fixed input geometry: the per-click forward is
    ~150 short launches at batch 2, i.e. launch-bound; replaying a graph removes the host from the loop.
    Inputs are copied into static buffers; `points` is padded with (-1,-1,-1) rows to a fixed capacity
    (ignored by the click-map kernel, so the result is identical).  Guidance-only upsampler work must be
    resident in the pointer-stable guidance cache before capture and is refreshed eagerly by the caller."""

    def __init__(self, net, image, points, capacity, token):
        B = image.shape[0]
        self.capacity = capacity
        self.image = image.clone()
        self.points = torch.full((B, 2 * capacity, 3), -1.0, device=image.device, dtype=torch.float32)
        self._load_points(points)
        side = torch.cuda.Stream(device=image.device)
        side.wait_stream(torch.cuda.current_stream(image.device))
        with torch.cuda.stream(side), guidance_scope(token):  # warm-up: attributes, packed weights, caches
            for _ in range(2):
                net(self.image, self.points)
        torch.cuda.current_stream(image.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), guidance_scope(token):
            self.out = net(self.image, self.points)["instances"]

    def _load_points(self, points):
        P = points.shape[1] // 2
        self.points.fill_(-1.0)
        self.points[:, :P] = points[:, :P]
        self.points[:, self.capacity:self.capacity + P] = points[:, P:]

    def replay(self, image, points):
        self.image.copy_(image)
        self._load_points(points)
        self.graph.replay()
        return self.out


class BasePredictor(object):
    def __init__(self, model, device: torch.device, net_clicks_limit: int = None, with_flip: bool = False,
                 zoom_in: BaseTransform = None, max_size: int = None, hip_graphs: bool = None, **kwargs) -> None:
        """``hip_graphs`` (default: env ISEGPROBE_HIP_GRAPHS) replays the per-click network call from a captured
        HIP graph; results are identical to the eager path (tests/test_inference_gpu.py)."""
        import os
        self.hip_graphs = bool(int(os.environ.get("ISEGPROBE_HIP_GRAPHS", "0"))) if hip_graphs is None else hip_graphs
        self._graphs = {}
        self.with_flip = with_flip
        self.net_clicks_limit = net_clicks_limit
        self.original_image = None
        self.device = device
        self.zoom_in = zoom_in
        self.prev_prediction = None
        self.model_indx = 0
        self.click_models = None
        self.net_state_dict = None
        if isinstance(model, tuple):
            self.net, self.click_models = model
        else:
            self.net = model
        self.transforms = [zoom_in] if zoom_in is not None else []
        if max_size is not None:
            self.transforms.append(LimitLongestSide(max_size=max_size))
        self.transforms.append(SigmoidForPred())
        if with_flip:
            self.transforms.append(AddHorizontalFlip())

    @staticmethod
    def to_tensor(image: np.ndarray) -> torch.Tensor:
        """torchvision ToTensor semantics: HWC uint8 -> CHW float in [0,1]."""
        image = np.ascontiguousarray(image)
        t = torch.from_numpy(image if image.flags.writeable else image.copy())  # (read-only arrays: decoded image buffers)
        if t.dim() == 2:
            t = t[:, :, None]
        t = t.permute(2, 0, 1)
        return t.float().div(255) if t.dtype == torch.uint8 else t.float()

    def set_input_image(self, image: Union[torch.Tensor, np.ndarray], **kwargs) -> None:
        image_nd = image if isinstance(image, torch.Tensor) else self.to_tensor(image)
        for transform in self.transforms:
            transform.reset()
        self.original_image = image_nd.to(self.device)
        if len(self.original_image.shape) == 3:
            self.original_image = self.original_image.unsqueeze(0)
        self.prev_prediction = torch.zeros_like(self.original_image[:, :1, :, :])
        self._guidance_token = object()  # new image: guidance-only upsampler work must be redone
        for net in (self.click_models or [self.net]):
            backbone = getattr(net, "backbone", None)
            if hasattr(backbone, "new_image"):
                backbone.new_image()  # activation ranges depend on the image: the half-precision trunk re-checks its range

    def _select_click_model(self, clicker, clicks_list):
        if self.click_models is not None:
            model_indx = min(clicker.click_indx_offset + len(clicks_list), len(self.click_models)) - 1
            if model_indx != self.model_indx:
                self.model_indx = model_indx
                self.net = self.click_models[model_indx]

    def get_prediction(self, clicker: Clicker, prev_mask: torch.Tensor = None) -> np.ndarray:
        return self.get_prediction_device(clicker, prev_mask).cpu().numpy()

    def get_prediction_device(self, clicker: Clicker, prev_mask: torch.Tensor = None) -> torch.Tensor:
        """get_prediction without the device->host copy: the [H0, W0] f32 probability map stays in HBM
        (consumed by DeviceClicker.evaluate_prediction)."""
        clicks_list = clicker.get_clicks()
        self._select_click_model(clicker, clicks_list)
        input_image = self.original_image
        if prev_mask is None:
            prev_mask = self.prev_prediction
        if hasattr(self.net, "with_prev_mask") and self.net.with_prev_mask:
            input_image = torch.cat((input_image, prev_mask), dim=1)
        image_nd, clicks_lists, is_image_changed = self.apply_transforms(input_image, [clicks_list])
        pred_logits = self._get_prediction(image_nd, clicks_lists, is_image_changed)
        prediction = self._resize_logits(pred_logits, image_nd.size()[2:])  # base_predictor.py:95-97
        prediction = self._inverse_transforms(prediction)
        if self.zoom_in is not None and self.zoom_in.check_possible_recalculation():
            return self.get_prediction_device(clicker)
        self.prev_prediction = prediction
        return prediction[0, 0]

    @staticmethod
    def _resize_logits(logits, size):
        if tuple(logits.shape[2:]) == tuple(size):
            return logits  # align_corners resize to the same size is the identity
        return ops.resize_bilinear_nchw_f32(logits.float().contiguous(), size[0], size[1])

    def _inverse_transforms(self, prediction):
        """Inverse chain in reversed order (base_predictor.py:99-102).  The common tail
        [.., SigmoidForPred, AddHorizontalFlip] is one fused kernel: sigmoid(0.5*(a + flip(b)))."""
        ts = list(reversed(self.transforms))
        if len(ts) >= 2 and isinstance(ts[0], AddHorizontalFlip) and isinstance(ts[1], SigmoidForPred):
            prediction = AddHorizontalFlip.fused_with_sigmoid(prediction)
            ts = ts[2:]
        for t in ts:
            prediction = t.inv_transform(prediction)
        return prediction

    def get_lowres_highres_feats(self, clicker: Clicker, prev_mask: torch.Tensor = None) -> Tuple[Dict, Dict]:
        if not hasattr(self.net, "get_lowres_highres_feats"):
            raise ValueError("Model does not support lowres-highres features extraction.")
        clicks_list = clicker.get_clicks()
        self._select_click_model(clicker, clicks_list)
        input_image = self.original_image.clone()
        if prev_mask is None:
            prev_mask = self.prev_prediction
        if hasattr(self.net, "with_prev_mask") and self.net.with_prev_mask:
            input_image = torch.cat((input_image, prev_mask), dim=1)
        image_nd, clicks_lists, _ = self.apply_transforms(input_image, [clicks_list])
        return self.net.get_lowres_highres_feats(image_nd, self.get_points_nd(clicks_lists))

    def _get_prediction(self, image_nd, clicks_lists, is_image_changed):
        # The RGB planes the upsamplers see are a function of (input image, crops applied by the transforms);
        # while that tuple is unchanged their click-independent work is reused.
        token = (id(self), getattr(self, "_guidance_token", None),
                 tuple(getattr(t, "applied_roi", None) for t in self.transforms))
        points = self.get_points_nd(clicks_lists)
        if self.hip_graphs and image_nd.is_cuda:
            return self._graphed(image_nd, points, token)
        with guidance_scope(token):
            return self.net(image_nd, points)["instances"]

    def _graphed(self, image_nd, points, token):
        P = points.shape[1] // 2
        key = (id(self.net), tuple(image_nd.shape))
        g = self._graphs.get(key)
        stale = g is not None and g.epoch != storage_epoch()  # cache storage the graph points into has moved
        if g is None or stale or g.capacity < P:
            if g is None and len(self._graphs) >= 4:  # images of many sizes without zoom-in: bound the graph pool
                self._graphs.pop(next(iter(self._graphs)))
            capacity = max(24, 2 * P) if g is None else (max(2 * g.capacity, P) if g.capacity < P else g.capacity)
            self._graphs.pop(key, None)
            del g
            g = self._graphs[key] = _ClickGraph(self.net, image_nd, points, capacity, token)
            g.token, g.epoch = token, storage_epoch()
        elif g.token != token:
            # the guidance changed (new image / zoom-in ROI): one eager pass refreshes the click-independent
            # upsampler work INTO the cache's existing storage (the captured graph keeps pointing at it) and is
            # itself this click's result; if that pass had to move storage the epoch check re-captures next time
            g.token = token
            with guidance_scope(token):
                return self.net(image_nd, points)["instances"]
        return g.replay(image_nd, points).clone()

    def _batch_infer(self, batch_image_tensor, batch_clickers, prev_mask=None):
        if prev_mask is None:
            prev_mask = self.prev_prediction
        input_image = batch_image_tensor
        if hasattr(self.net, "with_prev_mask") and self.net.with_prev_mask:
            input_image = torch.cat((batch_image_tensor, prev_mask), dim=1)
        clicks_lists = [clicker.get_clicks() for clicker in batch_clickers]
        image_nd, clicks_lists, _ = self.apply_transforms(input_image, clicks_lists)
        pred_logits = self.net(image_nd, self.get_points_nd(clicks_lists))["instances"]
        prediction = self._inverse_transforms(self._resize_logits(pred_logits, image_nd.size()[2:]))
        self.prev_prediction = prediction
        return prediction.cpu().numpy()[:, 0]

    def _get_transform_states(self):
        return [x.get_state() for x in self.transforms]

    def _set_transform_states(self, states):
        assert len(states) == len(self.transforms)
        for state, transform in zip(states, self.transforms):
            transform.set_state(state)

    def apply_transforms(self, image_nd: torch.Tensor, clicks_lists: List[List[Click]]) -> Tuple:
        is_image_changed = False
        for t in self.transforms:
            image_nd, clicks_lists = t.transform(image_nd, clicks_lists)
            is_image_changed |= t.image_changed
        return image_nd, clicks_lists, is_image_changed

    def get_points_nd(self, clicks_lists: List[List[Click]]) -> torch.Tensor:
        """[B, 2P, 3] (row, col, order), positives first, (-1,-1,-1) padding (base_predictor.py:194-225)."""
        num_pos = [sum(c.is_positive for c in clicks) for clicks in clicks_lists]
        num_neg = [len(clicks) - p for clicks, p in zip(clicks_lists, num_pos)]
        num_max_points = max(num_pos + num_neg)
        if self.net_clicks_limit is not None:
            num_max_points = min(self.net_clicks_limit, num_max_points)
        num_max_points = max(1, num_max_points)
        total = []
        for clicks in clicks_lists:
            clicks = clicks[: self.net_clicks_limit]
            pos = [c.coords_and_indx for c in clicks if c.is_positive]
            neg = [c.coords_and_indx for c in clicks if not c.is_positive]
            pos += (num_max_points - len(pos)) * [(-1, -1, -1)]
            neg += (num_max_points - len(neg)) * [(-1, -1, -1)]
            total.append(pos + neg)
        return torch.tensor(total, device=self.device, dtype=torch.float32)

    def get_states(self) -> Dict:
        return {"transform_states": self._get_transform_states(), "prev_prediction": self.prev_prediction.clone()}

    def set_states(self, states: Dict) -> None:
        self._set_transform_states(states["transform_states"])
        self.prev_prediction = states["prev_prediction"]
