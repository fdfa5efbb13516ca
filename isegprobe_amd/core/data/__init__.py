"""Train-time data feed of the probe (reference core/data/: base_dataset.py, points_sampler.py, datasets/sbd.py,
transforms.py).  Host-side: it produces the {"images", "instances", "points"} batches the HIP train step consumes
(core/training/trainer.py); nothing here is on the per-click dense-feature path."""
from .points_sampler import MultiPointSampler
from .sbd_train import SBDTrainSet, ShardSampler, make_loader
from .transforms import TrainAugmentor

__all__ = ["MultiPointSampler", "SBDTrainSet", "ShardSampler", "TrainAugmentor", "make_loader"]
