"""DistMaps / BatchImageNormalize / ScaleLayer with the reference's signatures
(core/model/ops.py:8-105), computed by HIP kernels."""
import torch
from torch import nn

from ... import hip_ops as ops


class DistMaps(nn.Module):
    """Click -> 2-channel disk / tanh-distance maps (ops.py:8-80).

    ``cpu_mode=True`` selected the Cython BFS in the reference (clicks rounded,
    _get_dist_maps.pyx:31); here it selects the same semantics on the GPU kernel
    (``round_clicks``) -- there is no host path."""

    def __init__(self, norm_radius, spatial_scale=1.0, cpu_mode=False, use_disks=False):
        super().__init__()
        self.spatial_scale = spatial_scale
        self.norm_radius = norm_radius
        self.cpu_mode = cpu_mode
        self.use_disks = use_disks

    def get_coord_features(self, points, batchsize, rows, cols):
        if points.shape[0] != batchsize:
            raise ValueError(f"points batch {points.shape[0]} != image batch {batchsize}")
        if self.cpu_mode:
            # ops.py:24-26: the BFS normalises by 1.0 for disks, radius*scale otherwise and
            # does not scale the coordinates
            return ops.click_maps(points, rows, cols, self.norm_radius * self.spatial_scale, 1.0,
                                  self.use_disks, round_clicks=True)
        return ops.click_maps(points, rows, cols, self.norm_radius, self.spatial_scale, self.use_disks)

    def forward(self, x, coords):
        return self.get_coord_features(coords, x.shape[0], x.shape[2], x.shape[3])


class ScaleLayer(nn.Module):
    """ops.py:83-93 (only used by RITM-style ``use_rgb_conv`` models; tiny, kept in torch)."""

    def __init__(self, init_value=1.0, lr_mult=1):
        super().__init__()
        self.lr_mult = lr_mult
        self.scale = nn.Parameter(torch.full((1,), init_value / lr_mult, dtype=torch.float32))

    def forward(self, x):
        return x * torch.abs(self.scale * self.lr_mult)


class BatchImageNormalize:
    """ops.py:96-105."""

    def __init__(self, mean, std, dtype=torch.float):
        self.mean = torch.as_tensor(mean, dtype=dtype)[None, :, None, None]
        self.std = torch.as_tensor(std, dtype=dtype)[None, :, None, None]
        self._mean3 = [float(v) for v in mean]
        self._std3 = [float(v) for v in std]

    def __call__(self, tensor):
        out, _ = ops.normalize(tensor.contiguous(), self._mean3, self._std3, want_prev=False)
        return out

    def split_and_normalize(self, tensor):
        """One pass over the 4-channel input: (normalised image, prev mask)."""
        return ops.normalize(tensor.contiguous(), self._mean3, self._std3, want_prev=True)
