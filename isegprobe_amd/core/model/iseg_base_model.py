"""Base class for interactive segmentation models (reference core/model/iseg_base_model.py:12-117)."""
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from ... import hip_ops as ops
from .ops import BatchImageNormalize, DistMaps


class iSegBaseModel(nn.Module):
    def __init__(self, with_aux_output: bool = False, norm_radius: int = 5, use_disks: bool = False,
                 cpu_dist_maps: bool = False, use_rgb_conv: bool = False, use_leaky_relu: bool = False,
                 with_prev_mask: bool = False,
                 norm_mean_std: Tuple[List, List] = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])) -> None:
        super().__init__()
        self.with_aux_output = with_aux_output
        self.with_prev_mask = with_prev_mask
        self.normalization = BatchImageNormalize(norm_mean_std[0], norm_mean_std[1])
        self.coord_feature_ch = 2 + (1 if with_prev_mask else 0)
        if use_rgb_conv:
            # RITM-only branch of the reference (iseg_base_model.py:39-59); not on the probed path
            raise NotImplementedError("use_rgb_conv (RITM) is outside the iSegProbe dense-feature path")
        self.maps_transform = nn.Identity()
        self.dist_maps = DistMaps(norm_radius=norm_radius, spatial_scale=1.0, cpu_mode=cpu_dist_maps,
                                  use_disks=use_disks)

    def forward(self, image: torch.Tensor, points: torch.Tensor) -> Dict:
        image, prev_mask = self.prepare_input(image)
        outputs = self._forward_prepared(image, prev_mask, points)
        outputs["instances"] = self._to_image_size(outputs["instances"], image.shape[2:])
        if self.with_aux_output:
            outputs["instances_aux"] = self._to_image_size(outputs["instances_aux"], image.shape[2:])
        return outputs

    def _forward_prepared(self, image, prev_mask, points):
        coord_features = self.get_coord_features(image, prev_mask, points)
        coord_features = self.maps_transform(coord_features)
        return self.backbone_forward(image, coord_features)

    @staticmethod
    def _to_image_size(logits, size):
        """F.interpolate(bilinear, align_corners=True) (iseg_base_model.py:75-80); with equal
        sizes the op is an exact identity, so the launch is skipped."""
        if tuple(logits.shape[2:]) == tuple(size):
            return logits
        if torch.is_grad_enabled() and logits.requires_grad:
            from ._autograd import ResizeLogitsFn
            return ResizeLogitsFn.apply(logits, size[0], size[1])
        return ops.resize_bilinear_nchw_f32(logits.float().contiguous(), size[0], size[1])

    def prepare_input(self, image: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.with_prev_mask:
            return self.normalization.split_and_normalize(image.float())
        return self.normalization(image.float()), None

    def backbone_forward(self, image, coord_features=None):
        raise NotImplementedError

    def get_coord_features(self, image, prev_mask, points):
        coord_features = self.dist_maps(image, points)
        if prev_mask is not None:
            coord_features = torch.cat((prev_mask, coord_features), dim=1)  # a device memcpy
        return coord_features

    def get_state_dict_to_save(self) -> Dict:
        return self.state_dict()
