"""DINO / timm ViT featurizer, the reference's "vit" backbone type (core/model/featurizers/DINO.py:
213-377, 470-611): a DINO-v1 style ViT-S/16 (no LayerScale, qkv bias, LayerNorm eps 1e-6) whose
dense features are either the LAST block's keys (``feat_type="key"``, channel = d*heads + head) or its
normalised patch tokens.  Same HIP engine as DINOv2Featurizer; for ``key`` the last block stops after
its QKV GEMM (attention and MLP of that block do not influence the keys).

The reference pulls weights from timm / torch.hub (DINO.py:497-510); here they come from ``weights=``
(state dict or path, DINO key layout) or ``$ISEGPROBE_DINO_WEIGHTS``; otherwise random init."""
import os

import torch

from .... import hip_ops as ops
from ...utils.log import logger
from .._autograd import TokenAddFn, TokenInjectFn, ViTTrunkFn
from .._tensor import BF16, nchw_view
from .DINOv2 import LN_EPS, DinoVisionTransformer, DINOv2Featurizer

ARCHS = {"vit_small": dict(embed_dim=384, depth=12, num_heads=6), "vit_base": dict(embed_dim=768, depth=12, num_heads=12)}


class DINOFeaturizer(DINOv2Featurizer):
    def __init__(self, arch: str, patch_size: int, feat_type: str = "key", feats_injection_mode: str = "no_injection",
                 weights=None, vit_kwargs=None) -> None:
        torch.nn.Module.__init__(self)
        self.arch = arch
        self.patch_size = patch_size
        self.feat_type = feat_type
        self.feats_injection_mode = feats_injection_mode
        assert feats_injection_mode in ["before_backbone", "after_backbone"], \
            f"Unknown feats_injection_mode: {feats_injection_mode}"  # DINO.py:517-520
        if feat_type not in ("key", "token"):
            raise ValueError("Unknown feat type:{}".format(feat_type))
        if vit_kwargs is None:
            key = "vit_base" if arch and "base" in arch else "vit_small"  # DINO.py:495 always builds vit_small
            vit_kwargs = dict(ARCHS[key], img_size=224)
        self.model = DinoVisionTransformer(patch_size=patch_size, init_values=None, mask_token=False, **vit_kwargs)
        self.n_feats = self.model.embed_dim
        weights = weights or os.environ.get("ISEGPROBE_DINO_WEIGHTS")
        if weights is not None:
            sd = torch.load(weights, map_location="cpu") if isinstance(weights, (str, os.PathLike)) else weights
            self.model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("head.")})
            logger.info(f"Loaded checkpoint for DINO: {arch}")
        else:
            logger.info(f"DINO {arch}: no weights given, keeping random init (no network for timm / torch.hub)")
        from .._tensor import PackedCache
        self._packed = PackedCache()
        self._pos_cache = {}

    def forward(self, img, additional_features=None, n: int = 1, include_cls: bool = False):
        if n != 1 or include_cls:
            raise NotImplementedError("only n=1 without the cls token is used by the probe")
        b, _, H, W = img.shape
        p = self.patch_size
        assert H % p == 0 and W % p == 0
        h, w = H // p, W // p
        if h % 2 == 1:
            raise AssertionError("odd patch-grid height: the reference's PatchEmbed drops a row and a column "
                                 "(DINO.py:205-208) and then fails its own reshape")
        D = self.model.embed_dim
        heads = self.model.num_heads
        mode = self.feats_injection_mode
        inject = additional_features is not None
        wants_grad = torch.is_grad_enabled() and inject and additional_features.requires_grad
        if wants_grad and mode == "before_backbone":  # the reference's training mode (models/sbd/vit/patch-embed_noup.py:43)
            if D // heads != 64:
                raise NotImplementedError("the attention backward is built for head_dim 64")
            with torch.no_grad():
                Wimg, bimg = self._image_weights()
                A = ops.patchify(img.float().contiguous(), None, None, p, Wimg.shape[1])
                xs, T = self._embed(A, Wimg, bimg, b, H, W)
            if tuple(additional_features.shape) != (b, T, D):
                raise AssertionError(f"x.shape: {(b, T, D)}, additional_features.shape: {tuple(additional_features.shape)}")
            x0 = TokenInjectFn.apply(xs, additional_features, b, T)
            feats = ViTTrunkFn.apply(x0, self.packed(), heads, b, T, LN_EPS, self.feat_type == "key")
            return nchw_view(feats.view(b, h, w, D))
        with torch.no_grad():
            Wimg, bimg = self._image_weights()
            A = ops.patchify(img.float().contiguous(), None, None, p, Wimg.shape[1])
            xs, T = self._embed(A, Wimg, bimg, b, H, W)
            if inject:
                if tuple(additional_features.shape) != (b, T, D):
                    raise AssertionError(f"x.shape: {(b, T, D)}, additional_features.shape: {tuple(additional_features.shape)}")
                if mode == "before_backbone":  # DINO.py:542-549
                    ops.token_add_(xs, additional_features, b, T, has_cls=True)
            if self.feat_type == "token":
                feats = self._blocks(xs, b, T)  # final norm, cls dropped
            else:
                qkv = self._blocks(xs, b, T, want_last_keys=True)
                k = qkv.view(b, T + 1, 3, heads, 64)[:, 1:, 1]             # [B,T,heads,64], cls removed (DINO.py:589)
                feats = k.permute(0, 1, 3, 2).reshape(b * T, D).contiguous()  # channel = d*heads + head (:590)
            if inject and mode == "after_backbone" and not wants_grad:  # DINO.py:572-580, :592-597
                ops.token_add_(feats, additional_features, b, T, has_cls=False)
        if wants_grad:
            feats = TokenAddFn.apply(feats.view(b, T, D), additional_features)
        return nchw_view(feats.view(b, h, w, D))
