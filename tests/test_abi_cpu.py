"""CPU: the C-ABI library loads and exports every symbol include/isegprobe_hip.h declares; the
product path refuses to run without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "isegprobe_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|long)\s+(isp_\w+)\s*\(", text, flags=re.M)))


def test_header_symbols_are_exported_and_bound():
    from isegprobe_amd import _lib
    names = _declared()
    assert len(names) >= 15
    handle = _lib.lib()
    for n in names:
        assert n in _lib.SIGNATURES, f"{n} declared in the header but missing from the binding table"
        assert hasattr(handle, n), f"{n} not exported by libisegprobe_hip.so"
    assert sorted(_lib.SIGNATURES) == names, "binding table lists symbols the header does not declare"
    assert handle.isp_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback():
    from isegprobe_amd import hip_ops
    from isegprobe_amd._lib import IspError
    with pytest.raises(IspError):
        hip_ops.layernorm(torch.zeros(4, 64), torch.ones(64), torch.zeros(64), 1e-6)
    with pytest.raises(IspError):
        hip_ops.conv3x3(torch.zeros(1, 4, 4, 64, dtype=torch.bfloat16), torch.zeros(64, 576, dtype=torch.bfloat16))


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "isegprobe_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f


def test_model_construction_and_state_dict_keys():
    from helpers import build_model
    m = build_model("bilinear")
    keys = set(m.state_dict())
    for k in ("backbone.model.cls_token", "backbone.model.pos_embed", "backbone.model.patch_embed.proj.weight",
              "backbone.model.blocks.0.attn.qkv.weight", "backbone.model.blocks.1.ls2.gamma",
              "backbone.model.norm.bias", "embed_coords.proj.weight", "head.convs.0.conv.weight",
              "head.convs.1.conv.bias", "head.classifier.weight"):
        assert k in keys, k
    saved = m.get_state_dict_to_save()
    assert set(saved) == keys  # no save_cfg -> everything
    m.save_cfg = {"backbone": False, "upsampler": False, "head": True, "embed_coords": True}
    assert all(k.startswith(("head.", "embed_coords.")) for k in m.get_state_dict_to_save())
    assert m._config["class"].endswith("iseg_probe_model.iSegProbeModel")


def test_serialize_roundtrip():
    from helpers import build_model
    from isegprobe_amd.core.utils.serialization import load_model
    m = build_model("identity")
    m2 = load_model(m._config)
    assert type(m2) is type(m) and set(m2.state_dict()) == set(m.state_dict())
    assert m2.upsampler_type == "identity" and m2.with_prev_mask


def test_checkpoint_roundtrip_reference_format(tmp_path):
    """{"state_dict", "config"} as core/utils/misc.py:68 writes it, resolved through the `core.*` alias
    exactly like a reference checkpoint (serialization.py:61-91)."""
    import isegprobe_amd
    from helpers import build_model
    from isegprobe_amd.core.utils.serialization import load_model
    m = build_model("bilinear")
    m.save_cfg = {"backbone": False, "upsampler": False, "head": True, "embed_coords": True}
    cfg = dict(m._config)
    cfg["class"] = "core.model.iseg_probe_model.iSegProbeModel"  # what a reference checkpoint names
    path = tmp_path / "last_checkpoint.pth"
    torch.save({"state_dict": m.get_state_dict_to_save(), "config": cfg}, str(path))
    isegprobe_amd.install_as_core()
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    m2 = load_model(ckpt["config"])
    msg = m2.load_state_dict(ckpt["state_dict"], strict=False)
    assert not msg.unexpected_keys and all(k.startswith(("backbone.", "upsampler.")) for k in msg.missing_keys)
    assert torch.equal(m2.head.convs[0].conv.weight, m.head.convs[0].conv.weight)
    assert type(m2).__name__ == "iSegProbeModel"


def test_grabcut_layout_reader(tmp_path):
    import numpy as np
    from isegprobe_amd.core.inference.datasets import GrabCutLayoutDataset, write_synthetic_grabcut
    ds = GrabCutLayoutDataset(write_synthetic_grabcut(tmp_path, n=3, size=(60, 80)))
    assert len(ds) == 3
    s = ds.get_sample(1)
    assert s.image.shape == (60, 80, 3) and s.image.dtype == np.uint8
    m = s.gt_mask(0)  # objects_ids are positions, as in the reference's DSample (data_sample.py:161-167)
    assert set(np.unique(m)) == {-1, 0, 1} and s.objects_ids == [0]


def test_save_checkpoint_and_load_is_model(tmp_path):
    """misc.save_checkpoint writes the reference's format (class path core.model..., trainable weights only under the
    model scripts' save_cfg); inference.utils.load_is_model rebuilds, loads, freezes it."""
    from helpers import build_model, seeded_
    from isegprobe_amd.core.inference.utils import load_is_model
    from isegprobe_amd.core.utils.misc import save_checkpoint
    m = seeded_(build_model("bilinear"), 4)
    m.save_cfg = {"embed_coords": True, "backbone": False, "upsampler": False, "head": True}
    path = save_checkpoint(m, tmp_path / "ckpts", verbose=False)
    assert path.name == "last_checkpoint.pth"
    assert save_checkpoint(m, tmp_path / "ckpts", epoch=7, prefix="run", verbose=False).name == "run_007.pth"
    ckpt = torch.load(str(path), map_location="cpu", weights_only=False)
    assert ckpt["config"]["class"] == "core.model.iseg_probe_model.iSegProbeModel"
    assert ckpt["state_dict"] and all(k.startswith(("head.", "embed_coords.")) for k in ckpt["state_dict"])
    m2 = load_is_model(str(path), torch.device("cpu"))
    assert torch.equal(m2.head.classifier.weight, m.head.classifier.weight)
    assert torch.equal(m2.embed_coords.proj.bias, m.embed_coords.proj.bias)
    assert not any(p.requires_grad for p in m2.parameters()) and not m2.training
