"""GPU: north_star's other named workload at its full size -- DINOv2-S/14 + LoftUp + ConvSegHead(384,2,1) at 448^2 (the
upsampler of BASELINE configs[2] / [4]; the `loftup448` block of bench.py), batch 8.  The CPU oracle needs tens of
seconds per image here, so the batch is checked through size-independent properties and tied to the oracle through one
of its images, as tests/test_fullsize_properties_gpu.py does for the FeatUp-JBU workload:

  * LoftUp's MinMaxScaler (loftup/layers.py:61-71) takes the colour minimum / maximum over the WHOLE BATCH, and its Fourier
    features multiply the scaled colours by frequencies up to e^10: the reference's output for an image depends on its batch
    mates, strongly (a 1e-5 change of the range moves the highest-frequency phases by 0.2 rad).  The images here carry an exact
    0 and an exact 1 in every channel, so every batch has the same range; then image i of the batch-8 forward equals the
    batch-1 forward of image i BIT FOR BIT (the pixel-map GEMMs' row tiles, the row statistics handed from the convolution /
    residual GEMMs to the LayerNorm-folded GEMMs, the (batch, head) cross-attention blocks and the conv tiles all fall on
    other workgroups in the two runs);
  * a permutation of the batch permutes the logits, bit for bit;
  * one image against the fp32 CPU oracle (both sides given that image alone) under the 16-bit gate (absolute 1e-2 on centred
    logits), masks equal away from the threshold."""
import numpy as np
import pytest
import torch

from helpers import S14, build_model, rand_points, seeded_

pytestmark = pytest.mark.gpu
B, SIZE, NPTS = 8, 448, 24


@pytest.fixture(scope="module")
def setup():
    model = build_model("loftup", vit=S14, img=(SIZE, SIZE), upsampler_params={"upsampler_path": None, "n_dim": 384})
    seeded_(model, 322)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    weights = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(6)
    image = torch.rand(B, 4, SIZE, SIZE)
    image[:, 3] = (image[:, 3] > 0.8).float()
    image[:, :3, 0, 0], image[:, :3, 0, 1] = 0.0, 1.0  # the same colour range in every image (see the module docstring)
    points = torch.from_numpy(rand_points(np.random.default_rng(6), B, NPTS, SIZE, SIZE))
    model = model.cuda()
    with torch.no_grad():
        full = model(image.cuda(), points.cuda())["instances"]
    return model, weights, image, points, full


def test_batch_entries_equal_single_image_runs(setup):
    model, _, image, points, full = setup
    for i in (0, 5):
        with torch.no_grad():
            one = model(image[i:i + 1].cuda(), points[i:i + 1].cuda())["instances"]
        assert torch.equal(one[0], full[i]), f"image {i}: max diff {(one[0] - full[i]).abs().max().item():.3g}"
    assert (full[0] - full[5]).abs().max().item() > 1e-2


def test_batch_permutation(setup):
    model, _, image, points, full = setup
    perm = torch.from_numpy(np.random.default_rng(8).permutation(B))
    with torch.no_grad():
        y = model(image[perm].cuda(), points[perm].cuda())["instances"]
    assert torch.equal(y, full[perm.cuda()])


def test_one_batch_entry_vs_oracle(setup):
    from oracle import model as omodel
    model, weights, image, points, full = setup
    i = 5
    cfg = dict(patch=14, depth=12, heads=6, upsampler="loftup", injection="before_backbone", with_prev_mask=True,
               use_disks=True, norm_radius=5)
    torch.set_num_threads(16)
    ref = omodel.forward(image[i:i + 1], points[i:i + 1], weights, cfg)[0]
    with torch.no_grad():
        y = model(image[i:i + 1].cuda(), points[i:i + 1].cuda())["instances"][0].cpu()
    shift = ref.median()
    err = (y - ref).abs()
    decided = (ref - shift).abs() > 1e-2
    agree = (((y - shift) > 0) == ((ref - shift) > 0))[decided].float().mean().item()
    print(f"LoftUp 448^2 image {i} vs oracle: max {err.max():.4g} rms {err.pow(2).mean().sqrt():.4g}, mask agreement {agree:.6f}")
    assert err.max().item() <= 1e-2 and agree == 1.0
