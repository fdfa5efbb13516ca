"""Train-time augmentation of an (image, instance map) pair (reference models/defaults.py:39-61: an albumentations pipeline
-- UniformRandomResize(0.75, 1.25), Flip, RandomRotate90, ShiftScaleRotate(+-3 degrees, p 0.75), PadIfNeeded(crop),
RandomCrop(crop), RandomBrightnessContrast, RGBShift; core/data/transforms.py:11-40 for the resize).  albumentations and
OpenCV are absent from the image; the same steps on numpy / PIL, with two stated differences: the +-3 degree / 3 % shift
jitter is left out, and resampling is PIL's bilinear (image) / nearest (instance ids) instead of OpenCV's."""
import random

import numpy as np
from PIL import Image


class TrainAugmentor:
    def __init__(self, crop_size, scale_range=(0.75, 1.25), brightness=(-0.25, 0.25), contrast=(-0.15, 0.4), rgb_shift=10,
                 p_color=0.75):
        self.crop = (int(crop_size[0]), int(crop_size[1]))
        self.scale_range, self.brightness, self.contrast = scale_range, brightness, contrast
        self.rgb_shift, self.p_color = rgb_shift, p_color

    def __call__(self, image: np.ndarray, mask: np.ndarray):
        """image [H, W, 3] uint8, mask [H, W] int32 instance ids -> the same, at exactly the crop size."""
        s = random.uniform(*self.scale_range)                      # UniformRandomResize
        H, W = mask.shape
        nh, nw = int(round(H * s)), int(round(W * s))
        image = np.asarray(Image.fromarray(image).resize((nw, nh), Image.BILINEAR))
        mask = np.asarray(Image.fromarray(mask.astype(np.int32)).resize((nw, nh), Image.NEAREST)).astype(np.int32)
        d = random.randint(-1, 1) if random.random() < 0.5 else None  # A.Flip: vertical, horizontal or both, p 0.5
        if d is not None:
            if d in (0, -1):
                image, mask = image[::-1], mask[::-1]
            if d in (1, -1):
                image, mask = image[:, ::-1], mask[:, ::-1]
        if random.random() < 0.5:                                   # A.RandomRotate90
            k = random.randint(0, 3)
            image, mask = np.rot90(image, k), np.rot90(mask, k)
        ch, cw = self.crop                                          # PadIfNeeded(border_mode=0) centred, then RandomCrop
        H, W = mask.shape
        ph, pw = max(ch - H, 0), max(cw - W, 0)
        if ph or pw:
            image = np.pad(image, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)))
            mask = np.pad(mask, ((ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)))
            H, W = mask.shape
        y0, x0 = random.randint(0, H - ch), random.randint(0, W - cw)
        image, mask = image[y0:y0 + ch, x0:x0 + cw], mask[y0:y0 + ch, x0:x0 + cw]
        img = image.astype(np.float32)
        if random.random() < self.p_color:                          # RandomBrightnessContrast (brightness_by_max)
            alpha, beta = 1.0 + random.uniform(*self.contrast), random.uniform(*self.brightness)
            img = img * alpha + beta * 255.0
        if random.random() < self.p_color:                          # RGBShift
            img = img + np.array([random.uniform(-self.rgb_shift, self.rgb_shift) for _ in range(3)], np.float32)
        return np.ascontiguousarray(np.clip(img, 0, 255).astype(np.uint8)), np.ascontiguousarray(mask)
