"""GPU: the dataset-level half of BASELINE.json's metric -- "NoC@90 parity on GrabCut" (configs[0]).

tests/golden/noc_dataset.npz holds what the REFERENCE's own GrabCutDataset + evaluate_dataset + compute_noc_metric
produced over the committed 50-image GrabCut-layout tree tests/golden/noc_grabcut/ (gen_golden.py::gen_noc_dataset:
NoBRS predictor, flip, zoom-in from the first click at the tiny model's 56 x 56, 20 clicks, thresh 0.5, every click run).
Here the same evaluation goes through THIS repo's evaluate.py (checkpoint -> load_is_model -> get_predictor ->
evaluate_dataset -> results table), on the HIP path."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, weights_from
from helpers import build_model

pytestmark = pytest.mark.gpu
THRS = (0.8, 0.85, 0.9)
# upsampler of the tiny model -> fixture.  "bilinear" is BASELINE configs[0]'s evaluation; the LoftUp / LiFT fixtures put the
# learned upsamplers of configs[2] / [3] through the same dataset-level evaluation (gen_golden.py::gen_noc_dataset_upsamplers)
FIXTURES = {"bilinear": "noc_dataset", "loftup": "noc_dataset_loftup", "lift": "noc_dataset_lift"}
UP_PARAMS = {"lift": {"lift_path": None, "n_dim": 128, "patch": 14}, "loftup": {"upsampler_path": None, "n_dim": 128}}
MIN_MID = {"bilinear": 26, "loftup": 15, "lift": 15}
FIXTURES = {up: name for up, name in FIXTURES.items() if os.path.exists(os.path.join(GOLDEN, name + ".npz"))}


def _checkpoint(golden, tmp_path, up):
    from isegprobe_amd.core.utils.misc import save_checkpoint
    g = golden(FIXTURES[up])
    model = build_model(up, upsampler_params=UP_PARAMS.get(up))
    missing, unexpected = model.load_state_dict(weights_from(g, "w"), strict=False)
    assert not unexpected and all(("mask_token" in k or "num_batches_tracked" in k) for k in missing), (missing, unexpected)
    return g, save_checkpoint(model, tmp_path / "ckpt", verbose=False)


def _run(golden, tmp_path, extra, up="bilinear"):
    import evaluate
    g, ckpt = _checkpoint(golden, tmp_path, up)
    res = evaluate.main(["--checkpoint", str(ckpt), "--dataset", os.path.join(GOLDEN, "noc_grabcut"), "--dataset-name", "GrabCut",
                         "--eval-mode", "fixed56", "--n-clicks", "20", "--thresh", "0.5", "--logs", str(tmp_path / "logs")] + extra)
    (name, all_ious, table), = res
    assert name == "GrabCut" and len(all_ious) == 50 and all(len(a) == 20 for a in all_ious)
    return g, np.stack(all_ious), table


def _noc(ious):
    return np.array([[(np.argmax(a >= t) + 1) if (a >= t).any() else 20 for t in THRS] for a in ious])


@pytest.mark.parametrize("up", list(FIXTURES))
def test_fixture_is_not_degenerate(golden, up):
    g = golden(FIXTURES[up])
    per = g["noc_per_object"][:, 2]
    assert ((per > 1) & (per < 20)).sum() >= MIN_MID[up]  # NoC@90 neither 1 nor 20 for most objects
    assert np.array_equal(_noc(g["ious"]), g["noc_per_object"])
    assert (g["clicks"][:, :, 2] == 0).any()              # negative clicks occur


@pytest.mark.parametrize("up,clicker", [(u, c) for u, c in (("bilinear", "device"), ("bilinear", "host"), ("lift", "device"), ("loftup", "device"))
                                        if u in FIXTURES])
def test_noc_dataset_identical_to_reference_fp32(golden, tmp_path, up, clicker):
    """Under the fp32-accurate forward the 50-object evaluation reproduces the reference's: NoC@80/85/90 per object and
    in the mean identical; the per-object IoU arrays identical wherever no prediction had a pixel within the fp32 gate of
    the threshold (|logit| < 1e-3; the generator stored those counts), and within a few pixels' worth of IoU elsewhere."""
    g, ious, table = _run(golden, tmp_path, ["--fp32"] + (["--host-clicker"] if clicker == "host" else []), up)
    ref = g["ious"]
    noc = _noc(ious)
    diff_obj = np.nonzero((noc != g["noc_per_object"]).any(1))[0]
    exact = np.array([np.array_equal(a, b) for a, b in zip(ious, ref)])
    clean = g["near_counts"].sum(1) == 0                  # objects none of whose 20 predictions had a near-threshold pixel
    print(f"[{up}, {clicker}] NoC@80/85/90 = {noc.mean(0)} (reference {g['noc']}); objects with identical IoU arrays {int(exact.sum())}/50 "
          f"({int(clean.sum())} have no near-threshold pixel at all); max |dIoU| {np.abs(ious - ref).max():.2e}; NoC differs on {diff_obj.tolist()}")
    assert exact[clean].all()
    assert np.abs(ious - ref).max() < 5e-3
    assert len(diff_obj) == 0, (diff_obj, noc[diff_obj], g["noc_per_object"][diff_obj])
    assert np.allclose(noc.mean(0), g["noc"]) and np.array_equal((noc == 20).sum(0), g["noc_over"])
    for k, t in enumerate(("NoC@80%", "NoC@85%", "NoC@90%")):   # the printed table row carries the same numbers
        assert abs(table[t] - g["noc"][k]) < 1e-12


@pytest.mark.parametrize("up", list(FIXTURES))
def test_noc_dataset_16bit_path(golden, tmp_path, up):
    """The product (16-bit operand) path on the same evaluation.  Logits carry <= 1e-2 of rounding noise, which moves mask
    pixels near the threshold and, through the robot user's argmax of a distance transform, sometimes a click.  Measured
    (MI355X, this fixture): NoC@80/85/90 5.80 / 8.22 / 12.54 against the reference's 5.82 / 8.22 / 12.26; 4 of 50 objects
    differ -- three are knife edges (the IoU sequences agree within 5e-3 at every click, and the reference's IoU sits within
    5e-3 of the threshold at the click where one side crosses it: object 10 hovers at 0.90 from click 6 on), one diverges
    at click 18 of 20.  Required: every differing (object, threshold) is one of those two kinds and is printed with its
    margin; mean NoC within half a click at every threshold; at most 8 objects differ; mean IoU within 5e-3.
    LiFT fixture (measured): 2.94 / 4.00 / 7.18 against 2.94 / 4.10 / 7.18, one knife-edge object; LoftUp fixture: 2.68 / 3.88 /
    5.84, identical to the reference's for every object."""
    g, ious, table = _run(golden, tmp_path, [], up)
    ref, noc = g["ious"], _noc(ious)
    thrs = (0.80, 0.85, 0.90)
    rows, unexplained = [], []
    for i in np.nonzero((noc != g["noc_per_object"]).any(1))[0]:
        far = np.abs(ious[i] - ref[i]) > 5e-3
        first = int(np.argmax(far)) + 1 if far.any() else 0   # first click at which the IoU sequence leaves the reference's
        for k, thr in enumerate(thrs):
            a, b = int(noc[i, k]), int(g["noc_per_object"][i, k])
            if a == b:
                continue
            c = min(a, b)                                      # the click at which one side reaches thr and the other does not
            margin = float(abs(ref[i, c - 1] - thr))
            kind = "knife-edge" if margin <= 5e-3 else ("diverged" if first and first <= c else "unexplained")
            rows.append((int(i), f"NoC@{int(thr * 100)}", a, b, f"ref IoU at click {c} is {margin:.1e} from the threshold", f"first |dIoU| > 5e-3 at click {first}", kind))
            if kind == "unexplained":
                unexplained.append(rows[-1])
    print(f"[{up}] NoC@80/85/90 = {noc.mean(0)} (reference {g['noc']}); mIoU@20 {ious[:, -1].mean():.4f} (reference {ref[:, -1].mean():.4f}); "
          f"{len(set(r[0] for r in rows))} objects differ (object, metric, here, reference, margin, divergence, kind):")
    for r in rows:
        print("   ", r)
    assert not unexplained, unexplained
    assert np.abs(noc.mean(0) - g["noc"]).max() <= 0.5
    assert len(set(r[0] for r in rows)) <= 8
    assert abs(ious.mean() - ref.mean()) < 5e-3


def test_clicks_limit_through_evaluate(golden, tmp_path):
    """eval_cfg.yaml `clicks_limit` (inference/utils.py:286-289): the network is fed at most 3 clicks of each polarity while
    the robot user keeps clicking.  The reference's own run of that setting (8 clicks per object, same tree and model) is in
    the fixture; here it goes through evaluate.py's --clicks-limit / --n-clicks in the fp32-accurate mode."""
    import evaluate
    g, ckpt = _checkpoint(golden, tmp_path, "bilinear")
    (name, all_ious, table), = evaluate.main(["--checkpoint", str(ckpt), "--dataset", os.path.join(GOLDEN, "noc_grabcut"), "--dataset-name", "GrabCut",
                                              "--eval-mode", "fixed56", "--n-clicks", "8", "--clicks-limit", "3", "--thresh", "0.5", "--fp32",
                                              "--logs", str(tmp_path / "logs")])
    ious, ref = np.stack(all_ious), g["limit3_ious"]
    assert ious.shape == ref.shape == (50, 8)
    exact = int(sum(np.array_equal(a, b) for a, b in zip(ious, ref)))
    print(f"clicks_limit=3: identical IoU arrays {exact}/50, max |dIoU| {np.abs(ious - ref).max():.2e}; "
          f"against the unlimited run the limit moves IoUs by up to {np.abs(ref - g['ious'][:, :8]).max():.2f}")
    assert np.abs(ious - ref).max() < 5e-3 and exact >= 45


def _run_sbd(golden, tmp_path, extra):
    import shutil
    import evaluate
    g = golden("noc_dataset_sbd")
    _, ckpt = _checkpoint(golden, tmp_path, "bilinear")
    tree = tmp_path / "noc_sbd"                      # (the SBD reader leaves its pair-list cache inside the tree)
    shutil.copytree(os.path.join(GOLDEN, "noc_sbd"), tree)
    (name, all_ious, table), = evaluate.main(["--checkpoint", str(ckpt), "--dataset", str(tree), "--dataset-name", "SBD", "--eval-mode", "fixed56",
                                              "--n-clicks", "20", "--thresh", "0.5", "--logs", str(tmp_path / "logs")] + extra)
    assert name == "SBD" and len(all_ious) == len(g["ious"]) == 46
    return g, np.stack(all_ious), table


def test_noc_sbd_layout_identical_to_reference_fp32(golden, tmp_path):
    """north_star: "NoC@90 on GrabCut/SBD identical to reference".  tests/golden/noc_sbd/ is a 26-image tree in SBD's on-disk
    layout (JPEG images, GTinst .mat instance maps, val.txt; two instances per image, no ignore band);
    noc_dataset_sbd.npz holds what the reference's SBDEvaluationDataset + evaluate_dataset + compute_noc_metric produced
    for its 46 (image, instance) objects with the bilinear NoC model.  fp32-accurate mode: NoC identical per object."""
    g, ious, table = _run_sbd(golden, tmp_path, ["--fp32"])
    ref, noc = g["ious"], _noc(ious)
    exact = int(sum(np.array_equal(a, b) for a, b in zip(ious, ref)))
    print(f"[SBD layout, fp32] NoC@80/85/90 = {noc.mean(0)} (reference {g['noc']}); identical IoU arrays {exact}/46; max |dIoU| {np.abs(ious - ref).max():.2e}")
    assert np.array_equal(noc, g["noc_per_object"]) and np.allclose(noc.mean(0), g["noc"])
    assert np.abs(ious - ref).max() < 5e-3
    for k, t in enumerate(("NoC@80%", "NoC@85%", "NoC@90%")):
        assert abs(table[t] - g["noc"][k]) < 1e-12


def test_noc_sbd_layout_16bit_path(golden, tmp_path):
    g, ious, table = _run_sbd(golden, tmp_path, [])
    ref, noc = g["ious"], _noc(ious)
    differ = np.nonzero((noc != g["noc_per_object"]).any(1))[0]
    print(f"[SBD layout, 16-bit] NoC@80/85/90 = {noc.mean(0)} (reference {g['noc']}); objects whose NoC differs: "
          f"{[(int(i), noc[i].tolist(), g['noc_per_object'][i].tolist()) for i in differ]}; mean |dIoU| {np.abs(ious - ref).mean():.2e}")
    assert np.abs(noc.mean(0) - g["noc"]).max() <= 0.5 and len(differ) <= 6
    assert abs(ious.mean() - ref.mean()) < 5e-3


def test_sharded_evaluation_over_two_ranks_equals_single_process(golden, tmp_path):
    """`torchrun --nproc-per-node 2 evaluate.py` shards the 50 images over the ranks (here both ranks on the box's
    one GPU, ISEGPROBE_SHARE_GPU=1): the IoU pickle rank 0 writes holds exactly the single-process run's arrays -- same order,
    same values -- and one table is printed."""
    import pickle
    import subprocess
    import sys
    import evaluate
    _, ckpt = _checkpoint(golden, tmp_path, "bilinear")
    args = ["--checkpoint", str(ckpt), "--dataset", os.path.join(GOLDEN, "noc_grabcut"), "--dataset-name", "GrabCut",
            "--eval-mode", "fixed56", "--n-clicks", "20", "--thresh", "0.5"]
    (_, ref, _), = evaluate.main(args + ["--logs", str(tmp_path / "logs1")])
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(ISEGPROBE_SHARE_GPU="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29500 + os.getpid() % 2000), os.path.join(root, "evaluate.py")] + args + ["--logs-path", str(tmp_path / "logs2")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.count("GrabCut: SPC") == 1  # rank 0's table only
    pk1, pk2 = sorted((tmp_path / "logs1").rglob("*.pkl")), sorted((tmp_path / "logs2").rglob("*.pkl"))
    assert pk2 and [p.name for p in pk1] == [p.name for p in pk2]
    for a, b in zip(pk1, pk2):  # <logs>/ious/GrabCut_fixed56_NoBRS_20.pkl: the list of per-object IoU arrays
        la, lb = pickle.load(open(a, "rb")), pickle.load(open(b, "rb"))
        assert len(la) == len(lb) == len(ref) == 50
        for x, y, z in zip(la, lb, ref):
            assert np.array_equal(np.asarray(x), np.asarray(y)) and np.array_equal(np.asarray(x), z)
    assert len({tuple(np.asarray(a).round(4)) for a in ref}) > 40  # the arrays differ between objects: the order is being tested


def test_save_feats_dump(golden, tmp_path):
    """evaluate.py --save-feats N (eval_cfg.yaml save_feats): LowRes / HighRes tensors of the first click of the first N images as
    fp32 NCHW .pth files plus the image with its click; the evaluation's IoUs are untouched by the extra forward."""
    import evaluate
    _, ckpt = _checkpoint(golden, tmp_path, "bilinear")
    args = ["--checkpoint", str(ckpt), "--dataset", os.path.join(GOLDEN, "noc_grabcut"), "--dataset-name", "GrabCut",
            "--eval-mode", "fixed56", "--n-clicks", "3", "--thresh", "0.5"]
    (_, plain, _), = evaluate.main(args + ["--logs", str(tmp_path / "a")])
    (_, dumped, _), = evaluate.main(args + ["--logs", str(tmp_path / "b"), "--save-feats", "2"])
    assert all(np.array_equal(x, y) for x, y in zip(plain, dumped))
    files = sorted(p.name for p in (tmp_path / "b" / "feats" / "GrabCut").rglob("*.pth"))
    assert files == ["0_0_HighRes.pth", "0_0_LowRes.pth", "1_0_HighRes.pth", "1_0_LowRes.pth"]
    root = next((tmp_path / "b" / "feats" / "GrabCut").iterdir())
    lo, hi = torch.load(root / "0_0_LowRes.pth"), torch.load(root / "0_0_HighRes.pth")
    assert lo.dtype == hi.dtype == torch.float32 and lo.is_contiguous() and hi.is_contiguous() and not lo.is_cuda
    assert lo.shape[0] == hi.shape[0] == 2 and lo.shape[1] == hi.shape[1]  # the flip pair; same channels
    assert lo.shape[2:] == (4, 4) and hi.shape[2:] == (56, 56) and torch.isfinite(hi).all() and float(hi.abs().max()) > 0
    assert sorted(p.name for p in (root / "images").iterdir()) == ["0_0_image.jpg", "1_0_image.jpg"]
