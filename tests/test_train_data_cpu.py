"""CPU: the train-time data feed and the epoch loop around the train step (reference core/data/points_sampler.py,
base_dataset.py, datasets/sbd.py, core/training/trainer.py:180-314) -- MultiPointSampler's click layout, the SBD train
set over the committed SBD-layout tree, per-rank shards, and two epochs x two steps at world size 2 on gloo with the
reference's LR milestones and checkpoint cadence (the HIP step itself: tests/test_training_gpu.py)."""
import os
import pickle
import random
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN

SBD = os.path.join(GOLDEN, "datasets", "sbd")


class _Scene:
    """Three rectangles on a 60 x 80 canvas as a DSample-like."""

    def __init__(self):
        self.mask = np.zeros((60, 80), np.int32)
        self.mask[5:30, 5:35] = 1
        self.mask[20:55, 45:75] = 2
        self.mask[40:56, 8:30] = 3
        self.ids = [1, 2, 3]

    def __len__(self):
        return len(self.ids)

    objects_ids = property(lambda self: list(range(len(self.ids))))

    def get_object_mask(self, i):
        return (self.mask == self.ids[i]).astype(np.int32)

    def get_background_mask(self):
        return self.mask == 0


def test_multi_point_sampler_layout_and_regions():
    from isegprobe_amd.core.data import MultiPointSampler
    from isegprobe_amd.core.data.points_sampler import generate_probs
    random.seed(1), np.random.seed(1)
    P = 8
    s = MultiPointSampler(P, prob_gamma=0.8, merge_objects_prob=0.5, max_num_merged_objects=2)
    scene = _Scene()
    n_pos, n_neg, merged = [], [], 0
    for _ in range(300):
        s.sample_object(scene)
        gt = s.selected_mask
        assert gt.shape == (1, 60, 80) and gt.dtype == np.float32 and set(np.unique(gt)) <= {0.0, 1.0}
        n_obj = sum(bool((gt[0] * (scene.mask == v)).sum()) for v in (1, 2, 3))
        assert n_obj in (1, 2)
        merged += n_obj == 2
        pts = np.array(s.sample_points(), np.float32)
        assert pts.shape == (2 * P, 3)
        pos, neg = pts[:P], pts[P:]
        for block, inside in ((pos, True), (neg, False)):
            valid = block[:, 0] >= 0
            assert np.all(block[~valid] == -1) and np.all(block[valid, 2] == 100)
            assert not valid[np.argmin(valid):].any() or valid.all()  # valid rows first, then the padding
            rc = block[valid, :2].astype(int)
            assert np.all(gt[0, rc[:, 0], rc[:, 1]] == (1.0 if inside else 0.0))
        assert (pos[:, 0] >= 0).sum() >= 1  # at least one positive click on a non-empty target
        n_pos.append(int((pos[:, 0] >= 0).sum())), n_neg.append(int((neg[:, 0] >= 0).sum()))
    assert 90 < merged < 210                      # merge_objects_prob = 0.5
    # a single object's positive count follows 1 + p(i) ~ 0.8^i; its mean over the draws is near the expectation
    assert abs(np.mean(n_neg) - float((np.arange(P + 1) * generate_probs(P + 1, 0.8)).sum())) < 0.6
    assert max(n_pos) <= P and max(n_neg) <= P
    # empty sample: zero target, no positives, negatives anywhere
    empty = _Scene()
    empty.mask[:] = 0
    empty.ids = []
    s.sample_object(empty)
    pts = np.array(s.sample_points())
    assert s.selected_mask.sum() == 0 and np.all(pts[:P] == -1)


def test_sbd_train_set_and_shards(tmp_path):
    from isegprobe_amd.core.data import SBDTrainSet, ShardSampler, make_loader
    random.seed(0), np.random.seed(0)
    ds = SBDTrainSet(SBD, crop_size=(56, 70), num_max_points=6, min_object_area=5, epoch_len=12)
    assert len(ds) == 12
    for i in range(6):
        b = ds[i]
        assert b["images"].shape == (3, 56, 70) and b["images"].dtype == torch.float32 and 0 <= b["images"].min() and b["images"].max() <= 1
        assert b["instances"].shape == (1, 56, 70) and set(b["instances"].unique().tolist()) <= {0.0, 1.0}
        assert b["points"].shape == (12, 3) and b["points"].dtype == torch.float32
        pos = b["points"][:6]
        rc = pos[pos[:, 0] >= 0, :2].long()
        assert torch.all(b["instances"][0, rc[:, 0], rc[:, 1]] == 1)
    # sampling weights (base_dataset.py:45-49, 109-122): rows (index, _, score); p ~ (1 - score)^gamma picks the index
    scores = tmp_path / "w.pkl"
    with open(scores, "wb") as f:
        pickle.dump([(0, None, 0.2)], f)
    dw = SBDTrainSet(SBD, crop_size=(56, 56), num_max_points=6, min_object_area=5, samples_scores_path=str(scores))
    assert dw.scores["indices"] == [0] and abs(dw.scores["probs"].sum() - 1) < 1e-12 and dw[0]["images"].shape == (3, 56, 56)
    # shards: every rank the same permutation per epoch, disjoint strided slices, a new permutation next epoch
    per_epoch = []
    for epoch in (0, 1):
        parts = []
        for r in range(2):
            sm = ShardSampler(11, seed=5, rank=r, world=2)
            sm.set_epoch(epoch)
            parts.append(list(sm))
            assert len(sm) == 6
        assert set(parts[0]) | set(parts[1]) == set(range(11)) and len(set(parts[0]) & set(parts[1])) <= 1
        per_epoch.append(parts)
    assert per_epoch[0] != per_epoch[1]
    ld = make_loader(ds, 3, workers=0, seed=0, rank=1, world=2)
    batches = list(ld)
    assert len(batches) == 2 and batches[0]["images"].shape == (3, 3, 56, 70) and batches[0]["points"].shape == (3, 12, 3)


def test_sbd_feed_identical_to_reference_sampler(golden):
    """The reference's own SBDDataset.__getitem__ + MultiPointSampler (tests/golden/gen_golden.py::gen_points_sampler: no
    augmentation, the SBD scripts' sampler settings with merge_objects_prob 0.5, seeded `random` / `numpy.random`) against this
    repo's feed under the same seeds: target masks and click lists identical draw for draw -- object choice, merging, erosion
    decisions, the counts from the gamma-decay distributions, the three negative regions and every coordinate."""
    from isegprobe_amd.core.data import MultiPointSampler, SBDTrainSet
    g = golden("points_sampler")
    for split in ("train", "val"):
        sampler = MultiPointSampler(6, prob_gamma=0.80, merge_objects_prob=0.5, max_num_merged_objects=2)
        ds = SBDTrainSet(SBD, split=split, min_object_area=20, keep_background_prob=0.01, points_sampler=sampler, augmentor=False)
        random.seed(123), np.random.seed(123)
        shape = tuple(g[f"{split}_shape"])
        for i in range(40):
            item = ds[0]
            assert tuple(item["instances"].shape) == shape
            assert np.array_equal(np.packbits(item["instances"][0].numpy() > 0), g[f"{split}_masks"][i]), (split, i)
            assert np.array_equal(item["points"].numpy(), g[f"{split}_points"][i]), (split, i, item["points"].numpy(), g[f"{split}_points"][i])
        assert np.array_equal((item["images"].numpy() * 255).round().astype(np.uint8), g[f"{split}_image"])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _CpuStepper:
    """The trainer's collective and optimiser plumbing (GradBucket with the head's early slice, Adam) around a surrogate
    loss on the REAL model's trainable parameters -- the HIP forward / backward needs the GPU."""

    def __init__(self, net):
        from isegprobe_amd.core.utils import distributed as D
        self.net = net
        params = [p for p in net.parameters() if p.requires_grad]
        head = [p for n, p in net.named_parameters() if n.startswith("head.") and p.requires_grad]
        self.bucket = D.GradBucket(params, early=head)
        self.optim = torch.optim.Adam(params, lr=5e-5)
        self.steps = []

    def step(self, batch):
        assert batch["images"].dim() == 4 and batch["points"].shape[1:] == (12, 3) and batch["instances"].shape[1] == 1
        self.bucket.zero()
        s = batch["images"].mean() + batch["instances"].mean() + (batch["points"][..., 0] >= 0).float().mean()
        loss = sum((p ** 2).mean() for p in self.bucket.params) * s
        self.bucket.arm_early()
        loss.backward()
        self.bucket.check_bound()
        self.bucket.finish_overlapped()
        self.optim.step()
        self.steps.append(float(s))
        return loss.detach()


def _worker(rank, world, port, ckpt_dir, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import hashlib
    import torch.distributed as dist
    from helpers import build_model, seeded_
    from isegprobe_amd.core.data import SBDTrainSet, make_loader
    from isegprobe_amd.core.training.trainer import EpochTrainer
    from isegprobe_amd.core.utils import distributed as D
    torch.set_num_threads(1)
    assert D.init_distributed("gloo")
    random.seed(rank), np.random.seed(rank)
    net = seeded_(build_model("bilinear"), 3)  # same weights on every rank
    net.save_cfg = {"embed_coords": True, "backbone": False, "upsampler": False, "head": True}
    ds = SBDTrainSet(SBD, crop_size=(56, 56), num_max_points=6, min_object_area=5, epoch_len=8)
    loader = make_loader(ds, 2, workers=0, seed=0)          # 8 samples / 2 ranks / batch 2 = 2 steps per epoch
    stepper = _CpuStepper(net)
    lrs = []
    et = EpochTrainer(stepper, loader, checkpoints_path=ckpt_dir, lr_milestones=[1], checkpoint_interval=[[0, 1]],
                      log=lambda m: lrs.append(m))
    hist = et.run(2)
    digest = hashlib.sha256(b"".join(p.detach().numpy().tobytes() for p in stepper.bucket.params)).hexdigest()
    gathered = [None] * world
    dist.all_gather_object(gathered, (digest, len(stepper.steps), stepper.steps, stepper.optim.param_groups[0]["lr"]))
    D.synchronize()
    if rank == 0:
        out.put((gathered, hist))
    dist.destroy_process_group()


def test_two_epochs_two_ranks_gloo(tmp_path):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    ckpt = str(tmp_path / "checkpoints")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ckpt, out)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, hist = out.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (d0, n0, s0, lr0), (d1, n1, s1, lr1) = gathered
    assert n0 == n1 == 4                                     # 2 epochs x 2 steps on each rank
    assert s0 != s1                                          # the ranks saw different samples ...
    assert d0 == d1                                          # ... and hold bit-identical parameters (averaged gradients)
    assert [h[0] for h in hist] == [0, 1] and abs(hist[0][2] - 5e-5) < 1e-12 and abs(hist[1][2] - 5e-6) < 1e-12  # milestone at epoch 1
    assert abs(lr0 - 5e-6) < 1e-12 and lr0 == lr1
    files = sorted(os.listdir(ckpt))
    assert files == ["000.pth", "001.pth", "last_checkpoint.pth"], files
    saved = torch.load(os.path.join(ckpt, "last_checkpoint.pth"), map_location="cpu", weights_only=False)
    keys = sorted(saved["state_dict"])
    assert "embed_coords.proj.weight" in keys and any(k.startswith("head.convs.0.conv") for k in keys) and "head.classifier.bias" in keys
    assert not any(k.startswith(("backbone.", "upsampler.")) for k in keys)
    assert saved["config"]["class"] == "core.model.iseg_probe_model.iSegProbeModel"
    # training.weights (trainer.py:550-557, 621-626): the probe checkpoint's tensors over a freshly built model
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import build_model, seeded_
    from isegprobe_amd.core.training.trainer import load_weights
    fresh = seeded_(build_model("bilinear"), 99)
    before = fresh.backbone.model.blocks[0].attn.qkv.weight.clone()
    msg = load_weights(fresh, os.path.join(ckpt, "last_checkpoint.pth"))
    assert not msg.unexpected_keys
    assert all(torch.equal(fresh.state_dict()[k], v) for k, v in saved["state_dict"].items())
    assert torch.equal(fresh.backbone.model.blocks[0].attn.qkv.weight, before)  # frozen parts are not in the file
