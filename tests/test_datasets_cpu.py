"""CPU: evaluation dataset readers against tiny on-disk fixtures written in each dataset's wire format
(reference core/data/datasets/*.py semantics: ignore labels, instance enumeration, caches)."""
import pickle

import numpy as np
import pytest
from PIL import Image
from scipy.io import savemat

from isegprobe_amd.core.inference import datasets as D


def _img(path, arr):
    Image.fromarray(arr).save(path)


def test_grabcut_layout_and_ignore_label(tmp_path):
    (tmp_path / "data_GT").mkdir()
    (tmp_path / "boundary_GT").mkdir()
    rgb = np.arange(6 * 8 * 3, dtype=np.uint8).reshape(6, 8, 3)
    m = np.zeros((6, 8), np.uint8)
    m[1:4, 2:6], m[4, 2:6] = 255, 128
    _img(tmp_path / "data_GT" / "a.png", rgb)
    _img(tmp_path / "boundary_GT" / "a.bmp", m)
    ds = D.get_dataset("GrabCut", tmp_path)
    s = ds.get_sample(0)
    assert len(ds) == 1 and s.objects_ids == [0] and np.array_equal(s.image, rgb)
    gt = s.gt_mask(0)
    assert gt.dtype == np.int32 and (gt == 1).sum() == 12 and (gt == -1).sum() == 4 and (gt == 0).sum() == 32


def test_davis_any_channel_is_object(tmp_path):
    (tmp_path / "img").mkdir()
    (tmp_path / "gt").mkdir()
    _img(tmp_path / "img" / "f0.jpg", np.full((5, 7, 3), 90, np.uint8))
    g = np.zeros((5, 7, 3), np.uint8)
    g[2, 3, 1], g[0, 0, 2] = 7, 200
    _img(tmp_path / "gt" / "f0.png", g)
    s = D.get_dataset("DAVIS", tmp_path).get_sample(0)
    gt = s.gt_mask(0)
    assert gt.sum() == 2 and gt[2, 3] == 1 and gt[0, 0] == 1 and s.image.shape == (5, 7, 3)


def _write_sbd(root, names):
    (root / "img").mkdir()
    (root / "inst").mkdir()
    for k, name in enumerate(names):
        _img(root / "img" / f"{name}.jpg", np.full((8, 9, 3), 40 * k, np.uint8))
        seg = np.zeros((8, 9), np.uint8)
        seg[0:3, 0:3] = 1
        seg[4:8, 2:9] = 2 + k
        seg[3, :] = 9 if k == 0 else 0      # a 1-pixel-high stripe across the image
        seg[7, 0] = 9 if k == 0 else 0      # ... plus one far pixel: bbox 5x9, area ratio 10/45 < 0.25
        savemat(root / "inst" / f"{name}.mat", {"GTinst": {"Segmentation": seg, "Categories": np.array([[1]])}})
    (root / "val.txt").write_text("\n".join(names) + "\n")
    (root / "train.txt").write_text(names[0] + "\n")


def test_sbd_evaluation_pairs_and_cache(tmp_path):
    _write_sbd(tmp_path, ["2008_a", "2008_b"])
    ds = D.get_dataset("SBD", tmp_path)
    assert ds.dataset_samples == [("2008_a", 1), ("2008_a", 2), ("2008_a", 9), ("2008_b", 1), ("2008_b", 3)]
    cache = tmp_path / "val_images_and_ids_list.pkl"
    assert cache.exists() and pickle.load(open(cache, "rb")) == ds.dataset_samples
    s = ds.get_sample(4)
    gt = s.gt_mask(0)
    assert s.objects_ids == [0] and gt.sum() == 28 and set(np.unique(gt)) == {0, 1}
    # an existing cache wins over the .mat files (reference behaviour): plant a shorter list
    pickle.dump([("2008_b", 1)], open(cache, "wb"))
    assert len(D.SBDEvaluationDataset(tmp_path, "val")) == 1
    assert len(D.get_dataset("SBD_Train", tmp_path)) == 3


def test_sbd_train_reader_drops_buggy_masks(tmp_path):
    _write_sbd(tmp_path, ["2008_a"])
    s = D.SBDDataset(tmp_path, "train", buggy_mask_thresh=0.25).get_sample(0)
    assert len(s) == 2  # instance 9 (area/bbox = 10/45) is removed, 1 and 2 stay
    assert s.gt_mask(0).sum() == 9 and s.gt_mask(1).sum() == 28
    assert len(D.SBDDataset(tmp_path, "train", buggy_mask_thresh=0.0).get_sample(0)) == 3


def test_pascal_test_split_grey_ids_and_void(tmp_path):
    for d in ("JPEGImages", "SegmentationObject", "ImageSets/Segmentation"):
        (tmp_path / d).mkdir(parents=True)
    _img(tmp_path / "JPEGImages" / "x.jpg", np.full((6, 6, 3), 128, np.uint8))
    pal = Image.new("P", (6, 6))
    pal.putpalette([0, 0, 0, 128, 0, 0, 0, 128, 0] + [0] * (253 * 3 - 3) + [224, 224, 192])  # VOC colours 0, 1, 2, 255
    px = np.zeros((6, 6), np.uint8)
    px[0:2, 0:3], px[3:6, 3:6], px[2, :] = 1, 2, 255
    pal.putdata(px.flatten().tolist())
    pal.save(tmp_path / "SegmentationObject" / "x.png")
    grey = {1: 38, 2: 75, 255: 220}  # OpenCV BGR2GRAY of (128,0,0), (0,128,0), (224,224,192)
    assert D._bgr2gray_u8(np.array([[[128, 0, 0], [0, 128, 0], [224, 224, 192]]], np.uint8)).tolist() == [[38, 75, 220]]
    pickle.dump((["x", "x"], [grey[1], grey[2]]), open(tmp_path / "ImageSets/Segmentation/test.pickle", "wb"))
    ds = D.get_dataset("PascalVOC", tmp_path)
    g0, g1 = ds.get_sample(0).gt_mask(0), ds.get_sample(1).gt_mask(0)
    assert (g0 == 1).sum() == 6 and (g1 == 1).sum() == 9 and (g0 == -1).sum() == 6 and (g1 == -1).sum() == 6


def test_unknown_dataset_name():
    with pytest.raises(NotImplementedError):
        D.get_dataset("Cityscapes", ".")


def test_results_table_and_log_files(tmp_path, capsys):
    """Wire format of the evaluation outputs (inference/utils.py:174-246,365-543): table columns, the log file that
    is appended to on later runs, the per-object IoU pickle and the plot pickle."""
    import pickle
    from isegprobe_amd.core.inference import utils as U
    ious = [np.array([0.5, 0.82, 0.91, 0.95], np.float32), np.array([0.3, 0.6, 0.7, 0.86], np.float32)]
    res = U.save_results("BilinearUpsampler", "GrabCut", tmp_path, (ious, 4.0), eval_mode="fixed224", n_clicks=4,
                         target_iou=1.01, print_ious=True, save_ious=True)
    out = capsys.readouterr().out.splitlines()
    head = [l for l in out if l.startswith("|   Upsampler Type")][0]
    assert [c.strip() for c in head.strip("|").split("|")] == ["Upsampler Type", "BRS Type", "Dataset", "NoC@80%", "NoC@85%",
                                                               "NoC@90%", "IoU@1", ">=4@85%", ">=4@90%", "SPC,s", "Time"]
    row = [l for l in out if l.startswith("| BilinearUpsampler")][0]
    cells = [c.strip() for c in row.split(";")[0].strip("|").split("|")]
    assert cells[:9] == ["BilinearUpsampler", "NoBRS", "GrabCut", "3.00", "3.50", "3.50", "0.40", "1", "1"]
    assert cells[9] == "0.500" and cells[10] == "0:00:04"
    assert "mIoU@1=40.00%;" in row and "mIoU@4=90.50%;" in row
    assert res["NoC@85%"] == 3.5 and res[">=4@90%"] == 1 and res["miou_list"][0] == 40.0 and res["clicks_list"] == [1, 2, 3, 4]
    log = tmp_path / "fixed224_NoBRS_4.txt"
    assert log.read_text().count("\n") == 5  # 4 header lines + 1 row
    U.save_results("BilinearUpsampler", "GrabCut", tmp_path, (ious, 4.0), eval_mode="fixed224", n_clicks=4, target_iou=1.01)
    assert log.read_text().count("\n") == 6  # appended, header not repeated
    saved = pickle.load(open(tmp_path / "ious" / "GrabCut_fixed224_NoBRS_4.pkl", "rb"))
    assert len(saved) == 2 and np.array_equal(saved[1], ious[1])
    p = U.save_iou_analysis_data("GrabCut", tmp_path, (ious, 4.0), eval_mode="fixed224", n_clicks=4)
    d = pickle.load(open(p, "rb"))
    assert p.name == "GrabCut_fixed224_NoBRS_4.pickle" and set(d) == {"dataset_name", "model_name", "all_ious"}
    assert d["model_name"].endswith("_NoBRS")


# ------------------------------------------------------------------ against the REFERENCE's own readers
@pytest.mark.parametrize("name,sub", [("GrabCut", "grabcut"), ("Berkeley", "berkeley"), ("DAVIS", "davis"), ("COCO_MVal", "davis"),
                                      ("SBD", "sbd"), ("SBD_Train", "sbd"), ("PascalVOC", "voc"), ("SBDDataset_train", "sbd")])
def test_readers_vs_reference_readers(golden, tmp_path, name, sub):
    """tests/golden/datasets/ holds a tiny tree in each on-disk layout; tests/golden/datasets.npz holds what the REFERENCE's
    readers (core/data/datasets/*.py + DSample, run by gen_golden.py::gen_datasets) returned for it: sample count, images,
    object ids and every gt_mask -- ours must return the same, first without and (SBD) then with the pickle cache."""
    import os
    import shutil
    g = golden("datasets")
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "datasets", sub)
    root = tmp_path / sub
    shutil.copytree(src, root)  # SBD's reader writes its cache next to the data
    for attempt in range(2 if name == "SBD" else 1):
        ds = D.SBDDataset(root, split="train") if name == "SBDDataset_train" else D.get_dataset(name, root)
        assert len(ds) == int(g[name + "_len"])
        for i in range(len(ds)):
            s = ds.get_sample(i)
            assert np.array_equal(s.image, g[f"{name}_{i}_image"]), (name, i)
            assert [int(v) for v in s.objects_ids] == g[f"{name}_{i}_objects"].tolist()
            for j in range(len(s.objects_ids)):
                ref = g[f"{name}_{i}_gt{j}"]
                got = s.gt_mask(j)
                assert got.shape == ref.shape and np.array_equal(got.astype(np.int32), ref), (name, i, j)
    if name == "SBD":
        assert (root / "val_images_and_ids_list.pkl").exists()


def test_noc_tree_read_like_the_reference(golden):
    """tests/golden/noc_grabcut is the 50-image tree the dataset-level NoC fixture was generated on: this repo's reader must
    hand the evaluation what the reference's GrabCutDataset did (grabcut.py:12-42) -- same order, image bytes, and object /
    background / ignore pixel counts (recorded by gen_golden.py::gen_noc_dataset from the reference's DSample)."""
    import os
    from isegprobe_amd.core.inference.datasets import GrabCutDataset
    g = golden("noc_dataset")
    ds = GrabCutDataset(os.path.join(os.path.dirname(__file__), "golden", "noc_grabcut"))
    assert len(ds) == 50 and [n for n in ds.dataset_samples] == [str(n) for n in g["names"]]
    for i in range(50):
        smp = ds.get_sample(i)
        assert tuple(smp.image.shape) == tuple(g[f"shape_{i}"]) and smp.image.dtype == np.uint8
        assert int(smp.image.astype(np.int64).sum()) == int(g[f"image_sum_{i}"])
        gt = smp.gt_mask(0)
        assert [int((gt == v).sum()) for v in (-1, 0, 1)] == g[f"gt_counts_{i}"].tolist()
        assert smp.objects_ids == [0]


def test_noc_sbd_tree_read_like_the_reference(golden, tmp_path):
    """tests/golden/noc_sbd (SBD layout) through this repo's SBD evaluation reader: the same (image, instance) pairs in the same
    order, image bytes and object pixel counts as the reference's SBDEvaluationDataset recorded (sbd.py:79-131)."""
    import os
    import shutil
    from isegprobe_amd.core.inference.datasets import get_dataset
    g = golden("noc_dataset_sbd")
    tree = tmp_path / "noc_sbd"
    shutil.copytree(os.path.join(os.path.dirname(__file__), "golden", "noc_sbd"), tree)
    ds = get_dataset("SBD", str(tree))
    assert len(ds) == len(g["pairs"]) == 46
    for i in range(len(ds)):
        name, inst = ds.dataset_samples[i]
        assert [int(name.split("_")[1]), int(inst)] == g["pairs"][i].tolist()
        smp = ds.get_sample(i)
        assert int(smp.image.astype(np.int64).sum()) == int(g[f"image_sum_{i}"])
        assert int((smp.gt_mask(0) == 1).sum()) == int(g[f"gt_count_{i}"])
