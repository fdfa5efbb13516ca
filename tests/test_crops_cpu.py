"""Crops (sliding-window) transform against the reference's own outputs (tests/golden/crops.npz, written by
tests/golden/gen_golden.py::gen_crops from reference core/inference/transforms/crops.py:14-117)."""
import os

import numpy as np
import torch

from isegprobe_amd.core.inference.clicker import Click
from isegprobe_amd.core.inference.transforms import Crops
from isegprobe_amd.core.inference.transforms.crops import get_offsets

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "crops.npz"))


def test_window_offsets_vs_reference():
    ptr, flat = G["offsets_ptr"], G["offsets_flat"]
    for i, (L, c, ov) in enumerate(G["offsets_args"]):
        assert get_offsets(int(L), int(c), float(ov)) == flat[ptr[i]:ptr[i + 1]].tolist(), (L, c, ov)


def test_forward_inverse_vs_reference():
    for tag in ("pass", "exact", "two_by_three", "tall"):
        H, W, ch, cw, ov = G[f"{tag}_args"]
        image = torch.from_numpy(G[f"{tag}_image"])
        clicks = [Click(bool(p), (int(y), int(x)), indx=int(i)) for p, y, x, i in G[f"{tag}_clicks"]]
        t = Crops(crop_size=(int(ch), int(cw)), min_overlap=float(ov))
        crops, lists = t.transform(image, [clicks])
        assert torch.equal(crops, torch.from_numpy(G[f"{tag}_crops"]))
        got = np.array([[[c.is_positive, c.coords[0], c.coords[1], c.indx] for c in lst] for lst in lists], np.int64)
        assert np.array_equal(got, G[f"{tag}_crop_clicks"])
        merged = t.inv_transform(torch.from_numpy(G[f"{tag}_probs"]))
        assert np.array_equal(merged.numpy(), G[f"{tag}_merged"])
        # state round trip + reset (base_transform protocol)
        t2 = Crops(crop_size=(int(ch), int(cw)), min_overlap=float(ov))
        t2.set_state(t.get_state())
        assert torch.equal(t2.inv_transform(torch.from_numpy(G[f"{tag}_probs"])), merged)
        t.reset()
        assert t.get_state() == (None, None, None)
    assert G["pass_crops"].shape[0] == 1 and G["two_by_three"+"_crops"].shape[0] == 6 and G["tall_crops"].shape[0] > 2
