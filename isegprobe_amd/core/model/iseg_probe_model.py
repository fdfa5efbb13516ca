"""Interactive segmentation model for probing VFMs and upsamplers (reference
core/model/iseg_probe_model.py:16-258): backbone -> upsampler -> head, every stage a HIP
module behind the reference's plugin contracts."""
from typing import Dict, Tuple

import os

import torch
import torch.nn as nn

from ... import hip_ops as ops
from ..utils.log import logger
from ..utils.model_builder import ModelBuilder
from ..utils.serialization import serialize
from ._tensor import BF16, nchw_view, to_nhwc_bf16
from .featurizers import DINOv2Featurizer
from .featurizers.utils import PatchEmbed
from .heads import ConvSegHead
from .upsamplers import JBUFeatUpUpsampler
from .iseg_base_model import iSegBaseModel


class iSegProbeModel(iSegBaseModel):
    """Same constructor as the reference (iseg_probe_model.py:34-46)."""

    @serialize
    def __init__(self, backbone_cfg: Dict = None, head_cfg: Dict = None, embed_coords_cfg: Dict = None,
                 neck_cfg: Dict = None, upsampler_cfg: Dict = None, save_cfg: Dict = None,
                 architecture: str = "backbone_upsampler_head", model_builder: ModelBuilder = None,
                 **kwargs) -> None:
        super().__init__(**kwargs)
        self.save_cfg = save_cfg
        self.fold_upsampler_affine = True  # cross-plugin weight folding (JBU fix-up -> head conv1); exact algebra
        self.head_f16 = os.environ.get("ISEGPROBE_HEAD_F16", "1") != "0"  # f16 head convs behind the JBU stack
        self.architecture = architecture
        assert backbone_cfg is not None and head_cfg is not None and embed_coords_cfg is not None, \
            "backbone, head and embed_coords configurations must be provided"
        assert self.architecture in ["backbone_upsampler_head", "backbone_neck_head"], \
            f"Unknown architecture: {self.architecture}"
        self.embed_coords_type = embed_coords_cfg["type"]
        self.upsampler_type = upsampler_cfg["type"] if upsampler_cfg else None
        self.model_builder = model_builder if model_builder is not None else ModelBuilder()

        # frozen modules
        self.backbone = self.model_builder.load_featurizer(backbone_cfg["type"], backbone_cfg["params"], freeze=True)
        self.upsampler = (
            self.model_builder.load_upsampler(upsampler_cfg["type"], upsampler_cfg["params"], freeze=True)
            if upsampler_cfg else self.model_builder.load_upsampler("bilinear", None, freeze=True))
        # trainable modules
        self.neck = (self.model_builder.load_neck(neck_cfg["type"], neck_cfg["params"], freeze=False)
                     if neck_cfg else nn.Identity())
        self.head = self.model_builder.load_head(head_cfg["type"], head_cfg["params"], freeze=False)
        if self.embed_coords_type == "patchEmbed":
            self.embed_coords = PatchEmbed(
                img_size=embed_coords_cfg["params"]["img_size"],
                patch_size=embed_coords_cfg["params"]["patch_size"],
                in_chans=3 if self.with_prev_mask else 2,
                embed_dim=embed_coords_cfg["params"]["embed_dim"])
        elif self.embed_coords_type == "simple_vit":
            self.embed_coords = self.model_builder.load_featurizer("simple_vit", embed_coords_cfg["params"], freeze=False)
        else:
            raise ValueError(f"Unknown embed_coords_type: {self.embed_coords_type}")
        self._count_parameters()

    # ---------------------------------------------------------------- forward
    def _fusable(self):
        return (isinstance(self.backbone, DINOv2Featurizer) and isinstance(self.embed_coords, PatchEmbed)
                and self.backbone.feats_injection_mode == "before_backbone"
                and self.embed_coords.patch_size[0] == self.backbone.patch_size
                and isinstance(self.maps_transform, nn.Identity))

    def _forward_prepared(self, image, prev_mask, points):
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if self._fusable() and not training:
            # click maps go straight into the patch matrix: no torch.cat, no token round trip
            maps = self.dist_maps(image, points)
            records = self._jbu_records_side_stream(image)
            # FeatUp JBU and LoftUp (inference) work in IEEE half from their first kernel on: the trunk's final LayerNorm
            # writes half for them (one bf16 rounding less between the trunk and the upsampler)
            from .upsamplers import BilinearUpsampler, LoftUpUpsampler
            half_out = (self.architecture == "backbone_upsampler_head"
                        and isinstance(self.upsampler, (JBUFeatUpUpsampler, LoftUpUpsampler)) and not self.upsampler.training)
            # ... and so does the head's first convolution when it runs through the bilinear plugin's resize
            half_out = half_out or (self.architecture == "backbone_upsampler_head" and isinstance(self.upsampler, BilinearUpsampler)
                                    and self._head_through_resize(image, image.shape[2] // self.backbone.patch_size,
                                                                  image.shape[3] // self.backbone.patch_size))
            feats = self.backbone.forward_fused_clicks(image, prev_mask, maps, self.embed_coords, out_f16=half_out)
            return self._after_backbone(image, feats, records)
        return super()._forward_prepared(image, prev_mask, points)

    def backbone_forward(self, image: torch.Tensor, coord_features: torch.Tensor = None) -> Dict:
        coord_features = self.embed_coords(coord_features)
        backbone_features = self.backbone(image, coord_features)
        return self._after_backbone(image, backbone_features)

    def _jbu_folds(self):
        return (self.architecture == "backbone_upsampler_head" and self.fold_upsampler_affine
                and isinstance(self.upsampler, JBUFeatUpUpsampler) and isinstance(self.head, ConvSegHead)
                and self.head.num_layers >= 1)

    def _jbu_records_side_stream(self, image):
        """FeatUp-JBU's kernel records depend on the image only and their kernels are VALU-bound (jbu_kernels: 59 % VALU,
        no MFMA), the ViT's are MFMA / staging-bound: the records are computed on a second HIP stream while the
        featurizer runs.  Returns (records, stream) or None."""
        if not (self._jbu_folds() and image.is_cuda and os.environ.get("ISEGPROBE_JBU_SIDE_STREAM", "1") != "0"):
            return None
        if torch.cuda.is_current_stream_capturing():
            return None  # inside a HIP-graph capture the forward stays on one stream
        main = torch.cuda.current_stream()
        side = getattr(self, "_side_stream", None)
        if side is None or side.device != image.device:
            # ISEGPROBE_SIDE_PRIO: HIP stream priority of the records' stream (torch: lower number = higher priority; the
            # featurizer runs on the caller's stream at the default priority 0)
            side = self._side_stream = torch.cuda.Stream(device=image.device, priority=int(os.environ.get("ISEGPROBE_SIDE_PRIO", "0")))
        side.wait_stream(main)
        p = self.backbone.patch_size
        with torch.cuda.stream(side):
            recs = self.upsampler.upsampler.stage_records(image, image.shape[2] // p, image.shape[3] // p, image.shape[2:])
        for r in recs:
            r.record_stream(main)
        return recs, side

    def _after_backbone(self, image, backbone_features, jbu_records=None):
        if self.architecture == "backbone_upsampler_head":
            if (not (torch.is_grad_enabled() and backbone_features.requires_grad)
                    and not (torch.is_grad_enabled() and any(p.requires_grad for p in self.head.parameters()))
                    and self.fold_upsampler_affine and isinstance(self.upsampler, JBUFeatUpUpsampler)
                    and not self.upsampler.upsampler.dropout_active()  # (train mode: the stack's Dropout2d layers run)
                    and isinstance(self.head, ConvSegHead) and self.head.num_layers >= 1):
                # JBUStack ends with z = x + 0.1*conv1x1(x); that affine map commutes with the bilinear
                # resize and folds into the head's first conv: the 2.5 TFLOP 1x1 GEMM disappears
                # (the resize to the image size is fused into the last JBU stage when 512 -> 448-like sizes allow it)
                records = None
                if jbu_records is not None:
                    records, side = jbu_records
                    torch.cuda.current_stream().wait_stream(side)
                # the stack's maps are IEEE half; the head takes them as they are when its f16 convolutions apply and
                # the resize is fused into the last stage (the stand-alone resize kernel is bf16): three more mantissa
                # bits on the head's input, weights and hidden map -- the largest share of the bf16 path's logit error
                stack = self.upsampler.upsampler
                p = self.backbone.patch_size
                half = (self.head_f16 and self.head.convs[0].takes_f16()
                        and stack.up4.resize_fusable((image.shape[2] // p) << 4, (image.shape[3] // p) << 4,
                                                     image.shape[2], image.shape[3]))
                hr = stack.forward_stages(backbone_features, image, out_size=image.shape[2:], records=records,
                                          out_dtype=ops.F16 if half else BF16)
                if image.size()[2:] != hr.size()[2:]:
                    hr = nchw_view(ops.resize_nhwc(to_nhwc_bf16(hr), image.shape[2], image.shape[3], "bilinear"))
                Wf, bf, alpha = self.upsampler.upsampler.fixup_affine()
                return {"instances": self.head.forward_folded_affine(hr, Wf, bf, alpha), "instances_aux": None}
            from .upsamplers import BilinearUpsampler
            if isinstance(self.upsampler, BilinearUpsampler) and self._head_through_resize(image, *backbone_features.shape[2:], backbone_features):
                # the plugin IS a bilinear(align_corners=True) resize to the guidance size (basic_upsamplers.py:28-33): the head's
                # first convolution is taken through it (heads/conv_heads.py forward_of_bilinear), the [B,H,W,C] map never exists
                return {"instances": self.head.forward_of_bilinear(backbone_features, image.shape[2], image.shape[3]),
                        "instances_aux": None}
            backbone_features = self.upsampler(source=backbone_features, guidance=image)
            return {"instances": self._resize_and_head(image, backbone_features), "instances_aux": None}
        backbone_features = self.neck(backbone_features, guidance=image)
        return {"instances": self.head(backbone_features), "instances_aux": None}

    def _head_through_resize(self, image, h, w, feats=None):
        """True when resize-to-image-size + head can run as ``head.forward_of_bilinear`` (inference only).  ``feats`` None:
        the question is asked before the features exist (their channel count is the head's)."""
        head = self.head
        if not hasattr(head, "of_bilinear_applies") or not image.is_cuda:
            return False
        if feats is not None:
            return head.of_bilinear_applies(feats, image.shape[2], image.shape[3])
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            return False
        return head.of_bilinear_geometry(head.in_channels, h, w, image.shape[2], image.shape[3])

    def _resize_and_head(self, image, hr_features):
        if (self.upsampler_type != "identity" and image.size()[2:] != hr_features.size()[2:]
                and self._head_through_resize(image, *hr_features.shape[2:], hr_features)):
            # iseg_probe_model.py:120-129 + conv_heads.py:69-73 as one operator (LiFT's 2h x 2w map, any plugin whose output
            # is at least ~5.7x smaller than the image)
            return self.head.forward_of_bilinear(hr_features, image.shape[2], image.shape[3])
        if self.upsampler_type != "identity" and image.size()[2:] != hr_features.size()[2:]:
            # iseg_probe_model.py:120-129: bilinear(align_corners=True) to the image size
            x = to_nhwc_bf16(hr_features)
            if torch.is_grad_enabled() and x.requires_grad:
                from ._autograd import ResizeBilinearFn
                hr_features = nchw_view(ResizeBilinearFn.apply(x, image.shape[2], image.shape[3]))
            else:
                hr_features = nchw_view(ops.resize_nhwc(x, image.shape[2], image.shape[3], "bilinear"))
        return self.head(hr_features)

    def forward_fp32(self, image: torch.Tensor, points: torch.Tensor) -> Dict:
        """forward() with fp32-accurate arithmetic ("three bf16 products", core/model/precise.py): the checking mode
        behind north_star's "logits within 1e-3 fp32" gate.  DINOv2 + identity / bilinear + conv heads only."""
        from .precise import forward_fp32
        return forward_fp32(self, image, points)

    def get_lowres_highres_feats(self, image: torch.Tensor, points: torch.Tensor) -> Tuple:
        """Low / high resolution features for PCA dumps (iseg_probe_model.py:136-174)."""
        image, prev_mask = self.prepare_input(image)
        coord_features = self.maps_transform(self.get_coord_features(image, prev_mask, points))
        lr_feats = self.backbone(image, self.embed_coords(coord_features))
        hr_feats = self.upsampler(source=lr_feats, guidance=image)
        feats = {"LowRes": lr_feats, "HighRes": hr_feats}
        if self.upsampler.__class__.__name__ in ["IdentityUpsampler", "LiFTUpsampler"]:
            feats["HighRes"] = nchw_view(ops.resize_nhwc(to_nhwc_bf16(hr_feats), image.shape[2], image.shape[3],
                                                         "bilinear"))
        return {"coord_features": coord_features}, feats

    def _count_parameters(self):
        n = lambda m: sum(p.numel() for p in m.parameters())
        params_count = {
            "backbone (M)": round(n(self.backbone) / 1e6, 2),
            "head (M)": round(n(self.head) / 1e6, 2),
            "embed_coords (k)": round(n(self.embed_coords) / 1e3, 2),
            "neck (M)": round(n(self.neck) / 1e6, 2),
            "trainable (M)": round(sum(p.numel() for p in self.parameters() if p.requires_grad) / 1e6, 2),
            "total (M)": round(n(self) / 1e6, 2),
        }
        if isinstance(self.upsampler, nn.Module):
            params_count["upsampler (k)"] = round(n(self.upsampler) / 1e3, 2)
        logger.info(f"PARAMETERS COUNT: {params_count}")

    def get_state_dict_to_save(self):
        """state_dict filtered by ``save_cfg`` (iseg_probe_model.py:199-258): keys of save_cfg are
        sub-module names; True keeps, False drops, {'save': bool, 'exclude': [...]} refines."""
        state_dict = self.state_dict()
        if not self.save_cfg:
            return state_dict

        def keep(name):
            cfg = self.save_cfg
            for part in name.split("."):
                if not isinstance(cfg, dict):
                    break
                if part in cfg.get("exclude", ()):
                    return False
                cfg = cfg.get(part, None)
                if cfg is False:
                    return False
                if cfg is None:
                    return True
                if isinstance(cfg, dict) and not cfg.get("save", False):
                    return False
            return True

        return {k: v for k, v in state_dict.items() if keep(k)}
