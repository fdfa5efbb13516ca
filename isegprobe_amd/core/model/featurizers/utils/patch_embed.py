"""Click-map patch embedding (reference core/model/featurizers/utils/patch_embed.py:12-42):
a k=p, s=p convolution, run as patchify + one bf16 MFMA GEMM."""
from typing import Tuple

import torch
from torch import nn

from ..... import hip_ops as ops
from ..._autograd import PatchEmbedFn, grad_mode
from ..._tensor import BF16, PackedCache


def _pad64(k):
    return (k + 63) // 64 * 64


class PatchEmbed(nn.Module):
    """2D map to patch tokens: [B,C,H,W] -> [B, h*w, D] (fp32 tokens)."""

    def __init__(self, img_size: Tuple[int, int] = (224, 224), patch_size: Tuple[int, int] = (16, 16),
                 in_chans: int = 3, embed_dim: int = 768, norm_layer=None, flatten: bool = True) -> None:
        super().__init__()
        self.in_chans = in_chans
        self.img_size = img_size
        self.patch_size = patch_size
        self.grid_size = (img_size[0] // patch_size[0], img_size[1] // patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)  # parameter holder
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()
        if patch_size[0] != patch_size[1]:
            raise NotImplementedError("square patches only")
        if norm_layer is not None or not flatten:
            raise NotImplementedError("PatchEmbed: norm_layer / flatten=False are not on the probed path")
        self._packed = PackedCache()

    def packed(self):
        def build():
            w = self.proj.weight.detach().flatten(1)
            K = w.shape[1]
            wp = torch.zeros(w.shape[0], _pad64(K), device=w.device, dtype=BF16)
            wp[:, :K] = w.to(BF16)
            return wp, self.proj.bias.detach().float().contiguous()
        return self._packed.get((self.proj.weight, self.proj.bias), build)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, C, H, W = x.shape
        p = self.patch_size[0]
        if grad_mode(self.proj):
            return PatchEmbedFn.apply(x, self.proj.weight, self.proj.bias, p, _pad64(self.proj.weight[0].numel()))
        wp, bias = self.packed()
        x = x.float().contiguous()
        A = ops.patchify(x, None, None, p, wp.shape[1])
        tokens = ops.linear(A, wp, bias, None, out_dtype=torch.float32)
        return tokens.view(B, (H // p) * (W // p), -1)
