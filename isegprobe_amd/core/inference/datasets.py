"""Evaluation dataset readers and their wire formats (reference core/data/datasets/{grabcut,berkeley,davis,
sbd,pascalvoc}.py, core/data/data_sample.py, core/inference/utils.py:86-104).  Host-side I/O around the
per-click path: what `evaluate.py` iterates over.

* GrabCut:            ``data_GT/<name>.*`` images, ``boundary_GT/<name>.*`` masks with {0, 128 = ignore, > 128 = object};
                      Berkeley: the same reader on ``images/`` + ``masks/``.
* DAVIS / COCO_MVal:  ``img/<name>.*``, ``gt/<name>.*`` (any non-zero channel = object).
* SBD:                ``img/<name>.jpg``, ``inst/<name>.mat`` (MATLAB struct ``GTinst.Segmentation``), ``{split}.txt``;
                      the evaluation variant enumerates (image, instance id) pairs and caches the list in
                      ``{split}_images_and_ids_list.pkl`` exactly like the reference, so caches are interchangeable.
* PascalVOC (test):   ``JPEGImages``, ``SegmentationObject`` (palette PNGs read as colour and reduced to OpenCV's
                      8-bit grey value), ``ImageSets/Segmentation/test.pickle`` = (names, instance grey values);
                      grey 220 (the VOC "void" colour) is the ignore region.

OpenCV is not a dependency: images are decoded with PIL (same libjpeg / libpng pixels); BGR2GRAY is OpenCV's
fixed-point formula restated."""
import pickle as pkl
from pathlib import Path
from typing import List

import numpy as np
from PIL import Image


def _imread_rgb(path) -> np.ndarray:
    """cv2.cvtColor(cv2.imread(path), COLOR_BGR2RGB): always 3 channels, alpha dropped, palettes expanded."""
    return np.asarray(Image.open(path).convert("RGB"))


def _bgr2gray_u8(rgb: np.ndarray) -> np.ndarray:
    """OpenCV's 8-bit BGR2GRAY: (R*4899 + G*9617 + B*1868 + 8192) >> 14."""
    r, g, b = (rgb[..., i].astype(np.int32) for i in range(3))
    return ((r * 4899 + g * 9617 + b * 1868 + 8192) >> 14).astype(np.int32)


def get_labels_with_sizes(x: np.ndarray):
    """core/utils/misc.py:123-127."""
    sizes = np.bincount(x.flatten())
    labels = [int(v) for v in np.nonzero(sizes)[0] if v != 0]
    return labels, sizes[labels].tolist()


def get_bbox_from_mask(mask: np.ndarray):
    """core/utils/misc.py:71-77."""
    rows, cols = np.any(mask, axis=1), np.any(mask, axis=0)
    rmin, rmax = np.where(rows)[0][[0, -1]]
    cmin, cmax = np.where(cols)[0][[0, -1]]
    return rmin, rmax, cmin, cmax


class DSample:
    """The evaluation-facing part of core/data/data_sample.py:13-176: an image, an encoded instance mask, the mask
    values that are objects and the values that are ignore regions.  ``objects_ids`` are positions (0..n-1), as in
    the reference; ``gt_mask(i)`` is {0, 1, -1 = ignore} int32."""

    def __init__(self, image: np.ndarray, encoded_masks: np.ndarray, objects_ids: List = None, ignore_ids: List = None,
                 sample_id: int = None):
        self.image = image
        self.sample_id = sample_id
        self._encoded_masks = encoded_masks
        self._object_values = list(objects_ids or [])
        self._ignore_values = list(ignore_ids or [])

    @property
    def objects_ids(self) -> List[int]:
        return list(range(len(self._object_values)))

    def get_object_mask(self, obj_id: int) -> np.ndarray:
        obj_mask = (self._encoded_masks == self._object_values[obj_id]).astype(np.int32)
        for v in self._ignore_values:
            obj_mask[self._encoded_masks == v] = -1
        return obj_mask

    def gt_mask(self, object_id: int = 0) -> np.ndarray:
        return self.get_object_mask(self.objects_ids[object_id])

    def __len__(self):
        return len(self._object_values)


class _Base:
    dataset_samples: list = ()

    def __len__(self):
        return len(self.dataset_samples)


class GrabCutLayoutDataset(_Base):
    """grabcut.py:12-42 / berkeley.py (same layout)."""

    def __init__(self, dataset_path, images_dir_name="data_GT", masks_dir_name="boundary_GT"):
        root = Path(dataset_path)
        self.dataset_path = root
        self._images_path, self._insts_path = root / images_dir_name, root / masks_dir_name
        self.dataset_samples = [x.name for x in sorted(self._images_path.glob("*.*"))]
        self._masks_paths = {x.stem: x for x in self._insts_path.glob("*.*")}

    def get_sample(self, index) -> DSample:
        name = self.dataset_samples[index]
        image = _imread_rgb(self._images_path / name)
        m = _imread_rgb(self._masks_paths[name.split(".")[0]])[:, :, 0].astype(np.int32)  # grabcut.py:35
        m[m == 128] = -1                                                                   # grabcut.py:37
        m[m > 128] = 1                                                                     # grabcut.py:38
        return DSample(image, m, objects_ids=[1], ignore_ids=[-1], sample_id=index)


GrabCutDataset = GrabCutLayoutDataset


class BerkeleyDataset(GrabCutLayoutDataset):
    """berkeley.py:6-10: the GrabCut reader on ``images/`` + ``masks/`` directories."""

    def __init__(self, dataset_path, **kwargs):
        super().__init__(dataset_path, images_dir_name="images", masks_dir_name="masks", **kwargs)


class DavisDataset(_Base):
    """davis.py:12-44 (also used for COCO_MVal, inference/utils.py:99-100)."""

    def __init__(self, dataset_path, images_dir_name="img", masks_dir_name="gt"):
        root = Path(dataset_path)
        self.dataset_path = root
        self._images_path, self._insts_path = root / images_dir_name, root / masks_dir_name
        self.dataset_samples = [x.name for x in sorted(self._images_path.glob("*.*"))]
        self._masks_paths = {x.stem: x for x in self._insts_path.glob("*.*")}

    def get_sample(self, index) -> DSample:
        name = self.dataset_samples[index]
        image = _imread_rgb(self._images_path / name)
        m = np.max(_imread_rgb(self._masks_paths[name.split(".")[0]]).astype(np.int32), axis=2)  # davis.py:40
        m[m > 0] = 1
        return DSample(image, m, objects_ids=[1], sample_id=index)


def _sbd_instances(path) -> np.ndarray:
    from scipy.io import loadmat
    return loadmat(str(path))["GTinst"][0][0][0].astype(np.int32)  # sbd.py:45


class SBDEvaluationDataset(_Base):
    """sbd.py:80-131: one sample per (image, instance); the pair list is cached beside the dataset."""

    def __init__(self, dataset_path, split="val"):
        assert split in {"train", "val"}
        self.dataset_path = Path(dataset_path)
        self.dataset_split = split
        self._images_path, self._insts_path = self.dataset_path / "img", self.dataset_path / "inst"
        with open(self.dataset_path / f"{split}.txt") as f:
            self._image_names = [x.strip() for x in f.readlines()]
        self.dataset_samples = self.get_sbd_images_and_ids_list()

    def get_sample(self, index) -> DSample:
        image_name, instance_id = self.dataset_samples[index]
        image = _imread_rgb(self._images_path / f"{image_name}.jpg")
        m = _sbd_instances(self._insts_path / f"{image_name}.mat")
        m[m != instance_id] = 0
        m[m > 0] = 1
        return DSample(image, m, objects_ids=[1], sample_id=index)

    def get_sbd_images_and_ids_list(self):
        pkl_path = self.dataset_path / f"{self.dataset_split}_images_and_ids_list.pkl"
        if pkl_path.exists():
            with open(pkl_path, "rb") as fp:
                return pkl.load(fp)
        pairs = []
        for name in self._image_names:
            ids, _ = get_labels_with_sizes(_sbd_instances(self._insts_path / f"{name}.mat"))
            pairs.extend((name, i) for i in ids)
        with open(pkl_path, "wb") as fp:
            pkl.dump(pairs, fp)
        return pairs


class SBDDataset(_Base):
    """sbd.py:15-77 (training split reader): all instances of an image, thin 'buggy' masks removed."""

    def __init__(self, dataset_path, split="train", buggy_mask_thresh=0.08):
        assert split in {"train", "val"}
        self.dataset_path = Path(dataset_path)
        self._images_path, self._insts_path = self.dataset_path / "img", self.dataset_path / "inst"
        self._buggy_objects, self._buggy_mask_thresh = {}, buggy_mask_thresh
        with open(self.dataset_path / f"{split}.txt") as f:
            self.dataset_samples = [x.strip() for x in f.readlines()]

    def get_sample(self, index) -> DSample:
        name = self.dataset_samples[index]
        image = _imread_rgb(self._images_path / f"{name}.jpg")
        m = self.remove_buggy_masks(index, _sbd_instances(self._insts_path / f"{name}.mat"))
        ids, _ = get_labels_with_sizes(m)
        return DSample(image, m, objects_ids=ids, sample_id=index)

    def remove_buggy_masks(self, index, instances_mask):
        if self._buggy_mask_thresh > 0.0:
            buggy = self._buggy_objects.get(index)
            if buggy is None:
                buggy = []
                for obj_id in get_labels_with_sizes(instances_mask)[0]:
                    obj = instances_mask == obj_id
                    r0, r1, c0, c1 = get_bbox_from_mask(obj)
                    if obj.sum() / ((r1 - r0 + 1) * (c1 - c0 + 1)) < self._buggy_mask_thresh:
                        buggy.append(obj_id)
                self._buggy_objects[index] = buggy
            for obj_id in buggy:
                instances_mask[instances_mask == obj_id] = 0
        return instances_mask


class PascalVocDataset(_Base):
    """pascalvoc.py:13-66."""

    def __init__(self, dataset_path, split="test"):
        assert split in {"train", "val", "trainval", "test"}
        self.dataset_path = Path(dataset_path)
        self._images_path, self._insts_path = self.dataset_path / "JPEGImages", self.dataset_path / "SegmentationObject"
        self.dataset_split = split
        seg = self.dataset_path / "ImageSets" / "Segmentation"
        if split == "test":
            with open(seg / "test.pickle", "rb") as f:
                self.dataset_samples, self.instance_ids = pkl.load(f)
        else:
            with open(seg / f"{split}.txt") as f:
                self.dataset_samples = [x.strip() for x in f.readlines()]

    def get_sample(self, index) -> DSample:
        sid = self.dataset_samples[index]
        image = _imread_rgb(self._images_path / f"{sid}.jpg")
        m = _bgr2gray_u8(_imread_rgb(self._insts_path / f"{sid}.png"))
        if self.dataset_split == "test":
            inst = self.instance_ids[index]
            mask = np.zeros_like(m)
            mask[m == 220] = 220
            mask[m == inst] = 1
            return DSample(image, mask, objects_ids=[1], ignore_ids=[220], sample_id=index)
        ids = [int(v) for v in np.unique(m) if v not in (0, 220)]
        return DSample(image, m, objects_ids=ids, ignore_ids=[220], sample_id=index)


def get_dataset(dataset_name: str, dataset_path):
    """core/inference/utils.py:86-104, the path given directly instead of through the hydra config."""
    table = {"GrabCut": lambda p: GrabCutDataset(p), "Berkeley": lambda p: BerkeleyDataset(p),
             "DAVIS": lambda p: DavisDataset(p), "COCO_MVal": lambda p: DavisDataset(p),
             "SBD": lambda p: SBDEvaluationDataset(p), "SBD_Train": lambda p: SBDEvaluationDataset(p, split="train"),
             "PascalVOC": lambda p: PascalVocDataset(p, split="test")}
    if dataset_name not in table:
        raise NotImplementedError(f"Dataset key: {dataset_name} is not found.")
    return table[dataset_name](dataset_path)


def write_synthetic_grabcut(root, n=50, seed=0, size=(300, 400)):
    """a GrabCut-*layout* fixture of seeded ellipses (the real dataset is not
    available offline)."""
    root = Path(root)
    (root / "data_GT").mkdir(parents=True, exist_ok=True)
    (root / "boundary_GT").mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(seed)
    H, W = size
    yy, xx = np.mgrid[:H, :W]
    for i in range(n):
        cy, cx = rng.uniform(0.35, 0.65) * H, rng.uniform(0.35, 0.65) * W
        ry, rx = rng.uniform(0.12, 0.3) * H, rng.uniform(0.12, 0.3) * W
        d = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2
        obj, band = d <= 1.0, (d > 1.0) & (d <= 1.15)
        img = rng.uniform(0, 70, (H, W, 3)) + obj[..., None] * rng.uniform(90, 160, 3)
        mask = np.zeros((H, W), np.uint8)
        mask[obj], mask[band] = 255, 128
        Image.fromarray(img.clip(0, 255).astype(np.uint8)).save(root / "data_GT" / f"{i:03d}.png")
        Image.fromarray(mask).save(root / "boundary_GT" / f"{i:03d}.bmp")
    return root
