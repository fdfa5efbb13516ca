#!/usr/bin/env python3
"""Errors of the 16-bit product path against the reference-generated fixtures (tests/golden): what the 1e-2 gates must hold.
Prints max / rms per fixture model and per featurizer fixture."""
import logging
import sys
import numpy as np
logging.disable(logging.CRITICAL)
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import weights_from
from helpers import TINY_VIT, build_model

g = dict(np.load("tests/golden/model_tiny.npz"))
image, points = torch.from_numpy(g["image"]).cuda(), torch.from_numpy(g["points"]).cuda()
for up in ("bilinear", "identity", "bilinear_after", "lift", "loftup"):
    inj = "after_backbone" if up.endswith("_after") else "before_backbone"
    base = up.replace("_after", "")
    kw = {}
    if base == "lift":
        kw = dict(upsampler_params={"lift_path": None, "n_dim": 128, "patch": 14})
    if base == "loftup":
        kw = dict(upsampler_params={"upsampler_path": None, "n_dim": 128})
    model = build_model(base, inj, **kw)
    w = {**weights_from(g, "common_w"), **weights_from(g, base + "_w")}
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    model = model.cuda().eval()
    with torch.no_grad():
        y = model(image, points)["instances"].cpu()
    ref = torch.from_numpy(g[up + "_logits"])
    e = (y - ref).abs()
    print(f"model {up:15s}: max {e.max():.4g} rms {e.pow(2).mean().sqrt():.4g}  (ref range {ref.min():.2f}..{ref.max():.2f})")

from isegprobe_amd.core.model.featurizers import DINOv2Featurizer
gv = dict(np.load("tests/golden/vit_tiny.npz"))
for inj in ("before_backbone", "after_backbone", "no_injection"):
    for tag in ("sq", "rect", "native"):
        f = DINOv2Featurizer("custom", inj, vit_kwargs=TINY_VIT)
        f.model.load_state_dict(weights_from(gv, "w"), strict=False)
        f = f.cuda().eval()
        y = f(torch.from_numpy(gv[f"{inj}_{tag}_x"]).cuda(), torch.from_numpy(gv[f"{inj}_{tag}_clicks"]).cuda()).float().cpu()
        ref = torch.from_numpy(gv[f"{inj}_{tag}_y"])
        e = (y - ref).abs()
        print(f"featurizer {inj}/{tag}: max {e.max():.4g} rms {e.pow(2).mean().sqrt():.4g} ref max {ref.abs().max():.3g} -> rel {e.max() / max(1.0, ref.abs().max()):.4g}")
