"""GrabCut / Berkeley-layout reader for evaluation (reference core/data/datasets/grabcut.py:12-42):
``<root>/data_GT/<name>.{jpg,png,bmp}`` images and ``<root>/boundary_GT/<name>.{bmp,png}`` masks with
values {0, 128 = ignore, 255 = object}.  PIL replaces cv2 (same decoded pixels for PNG/BMP)."""
from pathlib import Path

import numpy as np
from PIL import Image


class Sample:
    def __init__(self, image, mask, sample_id):
        self.image = image
        self._mask = mask
        self.objects_ids = [1]
        self.sample_id = sample_id

    def gt_mask(self, object_id):
        return self._mask


class GrabCutLayoutDataset:
    def __init__(self, dataset_path, images_dir_name="data_GT", masks_dir_name="boundary_GT"):
        root = Path(dataset_path)
        self._images_path, self._insts_path = root / images_dir_name, root / masks_dir_name
        self.dataset_samples = [x.name for x in sorted(self._images_path.glob("*.*"))]
        self._masks_paths = {x.stem: x for x in self._insts_path.glob("*.*")}

    def __len__(self):
        return len(self.dataset_samples)

    def get_sample(self, index):
        name = self.dataset_samples[index]
        image = np.asarray(Image.open(self._images_path / name).convert("RGB"))
        m = np.asarray(Image.open(self._masks_paths[name.split(".")[0]]))
        m = (m[:, :, 0] if m.ndim == 3 else m).astype(np.int32)
        mask = np.zeros_like(m)
        mask[m == 128] = -1   # grabcut.py:37
        mask[m > 128] = 1     # grabcut.py:38
        return Sample(image, mask, index)


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
