"""SBD training set and its loader (reference core/data/base_dataset.py:15-122, core/data/datasets/sbd.py:15-77,
models/defaults.py:81-90, core/training/trainer.py:60-92).

A sample = read an image and its instance map (core/inference/datasets.py::SBDDataset: ``img/*.jpg``, ``inst/*.mat``
``GTinst``, thin "buggy" masks dropped) -> augment until an object survives (or, with keep_background_prob, keep an empty
crop) -> drop objects below min_object_area -> MultiPointSampler picks the target and the clicks.  With a samples-scores
pickle (``assets/sbd_samples_weights.pkl``: [(index, ?, score), ...]) the index handed in by the sampler is ignored and
images are drawn with probability ~ (1 - score)^gamma, as base_dataset.py:45-49 does."""
import pickle
import random
from typing import Dict, Iterator

import numpy as np
import torch
from torch.utils import data

from ..inference.datasets import SBDDataset, get_labels_with_sizes
from ..utils import distributed as D
from .points_sampler import MultiPointSampler
from .transforms import TrainAugmentor


class _TrainSample:
    """What MultiPointSampler needs of core/data/data_sample.py:13-176 for a flat (one-layer) instance map."""

    def __init__(self, image, mask):
        self.image, self.mask = image, mask
        self.ids, self.areas = get_labels_with_sizes(mask)

    def remove_small_objects(self, min_area):
        """data_sample.py:127-133, 202-215: the object leaves the list, its pixels stay in the instance map -- they are not
        background afterwards, so they still count as "other objects" for the negative clicks (pinned by the sampler fixture)."""
        keep = [(i, a) for i, a in zip(self.ids, self.areas) if a >= min_area]
        self.ids, self.areas = [i for i, _ in keep], [a for _, a in keep]

    def __len__(self):
        return len(self.ids)

    @property
    def objects_ids(self):
        return list(range(len(self.ids)))

    def get_object_mask(self, obj):
        return (self.mask == self.ids[obj]).astype(np.int32)

    def get_background_mask(self):
        return self.mask == 0


class SBDTrainSet(data.Dataset):
    def __init__(self, dataset_path, crop_size=(224, 224), num_max_points=24, split="train", min_object_area=80,
                 keep_background_prob=0.01, samples_scores_path=None, samples_scores_gamma=1.25, augmentor=None,
                 points_sampler=None, epoch_len=-1):
        self.reader = SBDDataset(dataset_path, split=split)
        # augmentor: None = the SBD scripts' pipeline at `crop_size`; False = none (the reference's augmentator=None: samples as read)
        self.augmentor = TrainAugmentor(crop_size) if augmentor is None else augmentor
        self.points_sampler = points_sampler or MultiPointSampler(num_max_points, prob_gamma=0.80, merge_objects_prob=0.15,
                                                                 max_num_merged_objects=2)  # models/defaults.py:74-79
        self.min_object_area, self.keep_background_prob, self.epoch_len = min_object_area, keep_background_prob, epoch_len
        self.scores = self._load_scores(samples_scores_path, samples_scores_gamma)

    @staticmethod
    def _load_scores(path, gamma):
        """base_dataset.py:109-122."""
        if path is None:
            return None
        with open(path, "rb") as f:
            rows = pickle.load(f)
        probs = np.array([(1.0 - r[2]) ** gamma for r in rows], np.float64)
        return {"indices": [r[0] for r in rows], "probs": probs / probs.sum()}

    def __len__(self):
        return self.epoch_len if self.epoch_len > 0 else len(self.reader)

    def __getitem__(self, index) -> Dict:
        if self.scores is not None:
            index = int(np.random.choice(self.scores["indices"], p=self.scores["probs"]))
        elif self.epoch_len > 0:
            index = random.randrange(0, len(self.reader))
        raw = self.reader.get_sample(index)
        image0, mask0 = raw.image, raw._encoded_masks.astype(np.int32)
        while True:  # base_dataset.py:78-91: re-draw the augmentation until an object is left (or an empty crop is kept)
            image, mask = self.augmentor(image0, mask0.copy()) if self.augmentor else (image0, mask0.copy())
            s = _TrainSample(image, mask)
            if not self.augmentor:  # (base_dataset.py:79-80: without an augmentator the sample is returned as it is)
                break
            if len(s) > 0 or self.keep_background_prob < 0.0 or random.random() < self.keep_background_prob:
                break
        s.remove_small_objects(self.min_object_area)
        self.points_sampler.sample_object(s)
        points = np.array(self.points_sampler.sample_points(), np.float32)
        return {"images": torch.from_numpy(image.copy()).permute(2, 0, 1).float().div(255.0),  # transforms.ToTensor()
                "points": torch.from_numpy(points), "instances": torch.from_numpy(self.points_sampler.selected_mask.copy())}


class ShardSampler(data.Sampler):
    """This rank's share of an epoch: a seeded permutation of the dataset (seed + epoch, the same on every rank) cut into
    disjoint strided shards of equal length (utils/distributed.shard_indices) -- DistributedSampler's contract
    (trainer.py:60-68; ``set_epoch`` as at trainer.py:204-205)."""

    def __init__(self, n, shuffle=True, seed=0, rank=None, world=None):
        self.n, self.shuffle, self.seed, self.epoch = n, shuffle, seed, 0
        self.rank, self.world = rank, world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def _indices(self):
        order = list(range(self.n))
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = torch.randperm(self.n, generator=g).tolist()
        return [order[i] for i in D.shard_indices(self.n, self.rank, self.world)]

    def __iter__(self) -> Iterator[int]:
        return iter(self._indices())

    def __len__(self):
        return len(D.shard_indices(self.n, self.rank, self.world))


def _seed_worker(worker_id):
    seed = torch.initial_seed() % 2 ** 31  # per worker and per epoch (torch derives it from the loader's base seed)
    np.random.seed(seed)
    random.seed(seed)


def make_loader(dataset, batch_size, workers=0, seed=0, rank=None, world=None, shuffle=True):
    """DataLoader over this rank's shard (trainer.py:60-78: batch_size per process = global // world, drop_last, pinned)."""
    sampler = ShardSampler(len(dataset), shuffle=shuffle, seed=seed, rank=rank, world=world)
    g = torch.Generator().manual_seed(seed * 1000 + (D.get_rank() if rank is None else rank))
    return data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, drop_last=True, num_workers=workers,
                           pin_memory=torch.cuda.is_available(), worker_init_fn=_seed_worker, generator=g)
