"""String -> class factory for featurizers / upsamplers / heads (reference
core/utils/model_builder.py:13-100); freezes parameters like the reference."""
from typing import Dict

import torch.nn as nn

from .log import logger

# the plugin registries are imported inside the loaders: `core.model` imports this module
# (iSegProbeModel's default builder), so a module-level import would be circular


class ModelBuilder:
    """Class to load different components of interactive segmentation model."""

    def __init__(self) -> None:
        pass

    def load_featurizer(self, type: str, params: Dict, freeze: bool = True) -> nn.Module:
        from ..model.featurizers import DINOFeaturizer, DINOv2Featurizer, MaskCLIPFeaturizer, SimpleViTFeaturizer
        type = type.lower()
        if type == "dinov2":
            backbone = DINOv2Featurizer(**params)
        elif type == "vit":
            backbone = DINOFeaturizer(**params)
        elif type == "simple_vit":  # model_builder.py:40-49
            backbone = SimpleViTFeaturizer(image_size=params["img_size"], patch_size=params["patch_size"],
                                           dim=params["embed_dim"], depth=params["depth"], heads=params["heads"],
                                           mlp_dim=params["mlp_dim"], channels=params["channels"],
                                           dim_head=params["dim_head"])
        elif type == "mask_clip":
            backbone = MaskCLIPFeaturizer(**params)
        else:
            raise ValueError(f"Unsupported backbone type: {type}")
        if freeze:
            for param in backbone.parameters():
                param.requires_grad = False
        return backbone

    def load_upsampler(self, type: str, params: Dict = None, freeze: bool = True):
        from ..model.upsamplers import UPSAMPLER_REGISTRY
        type = type.lower()
        if type not in UPSAMPLER_REGISTRY:
            raise ValueError(f"Unsupported upsampler type: {type}")
        upsampler_cls = UPSAMPLER_REGISTRY[type]
        upsampler = upsampler_cls(**params) if params else upsampler_cls()
        if freeze:
            for param in upsampler.parameters():
                param.requires_grad = False
        logger.info(f"UPSAMPLER: Loaded {upsampler.__class__.__name__}")
        return upsampler

    def load_head(self, type: str, params: Dict, freeze: Dict = False):
        from ..model.heads import HEAD_REGISTRY
        if type not in HEAD_REGISTRY:
            raise ValueError(f"Unsupported head type: {type}")
        head = HEAD_REGISTRY[type](**params)
        logger.info(f"HEAD: Loaded {head.__class__.__name__}")
        if freeze:
            for param in head.parameters():
                param.requires_grad = False
        return head

    def load_neck(self, type: str, params: Dict, freeze=False) -> nn.Module:
        raise NotImplementedError("Neck loading is not implemented yet. Please implement the neck loading logic.")
