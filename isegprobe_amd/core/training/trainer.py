"""Data-parallel training step for the probe (reference core/training/trainer.py:193-314,377-477,
577-618), reduced to what sits on the dense-feature path:

  batch -> [0..max_num_next_clicks no-grad forwards that simulate corrective clicks] -> forward
        -> NFL loss -> backward (HIP kernels through core/model/_autograd.py) -> flat-bucket all-reduce of the
        trainable gradients (RCCL over xGMI; utils/distributed.GradBucket: the head's slice is issued
        asynchronously inside backward, the click encoder's after it)
        -> Adam step.

The reference wraps the net in DistributedDataParallel; here the collective is explicit and sits
between backward and the optimizer step, on the gradient step only.  Dataset readers, augmentation,
samplers, logging and checkpoint cadence are outside this path.
Click simulation (get_next_points) runs on the device: OpenCV's 5x5 chamfer DIST_L2 transform (the
reference's cv2.distanceTransform(mask, DIST_L2, 5)) restated as a wavefront kernel, polarity choice and
uniform draw included -- no device->host copy per simulated click."""
import random
from typing import Dict

import numpy as np
import torch

from ... import hip_ops as ops
from ..model._guidance_cache import guidance_scope
from ..utils import distributed as D
from .losses import NormalizedFocalLossSigmoid


def get_next_points(pred, gt, points, click_indx, pred_thresh=0.49, rng=np.random, workspace=None):
    """Train-time click simulation (trainer.py:575-618) on the device: FN / FP masks of `pred`, OpenCV's 5x5
    chamfer DIST_L2 transform of the zero-padded masks, larger maximum picks the polarity, uniform draw among the
    pixels with dt > max/2.  Nothing is copied to the host; the only host work is drawing one 32-bit integer per
    sample from `rng` (where the reference calls np.random.randint(0, n) once it knows n)."""
    if not pred.is_cuda:
        raise RuntimeError("get_next_points runs on the GPU (no CPU fallback)")
    draws = torch.from_numpy(rng.randint(0, 2 ** 32, size=pred.shape[0], dtype=np.int64))
    out, _ = ops.next_points(pred, gt, points, click_indx, draws, pred_thresh, workspace)
    return out


class DataParallelTrainer:
    def __init__(self, model, lr=5e-5, betas=(0.9, 0.999), eps=1e-8, max_num_next_clicks=3,
                 prev_mask_drop_prob=0.0, loss=None, frozen_bn_batch_stats=True):
        """``frozen_bn_batch_stats`` (default, = the reference): ``net.train()`` (trainer.py:214,431) also puts the
        BatchNorm2d layers of the frozen LiFT / LoftUp upsamplers into batch-statistics mode (forward, backward and
        running-statistics update).  False keeps the frozen upsampler in eval mode during the training forward."""
        self.net = model
        self.frozen_bn_batch_stats = frozen_bn_batch_stats
        self.loss_fn = loss or NormalizedFocalLossSigmoid(alpha=0.5, gamma=2)
        self.max_num_next_clicks = max_num_next_clicks
        self.prev_mask_drop_prob = prev_mask_drop_prob
        params = [p for p in model.parameters() if p.requires_grad]
        # the head's gradients are final before the upsampler / trunk data gradients start: their slice of the bucket is
        # reduced while the rest of backward runs (GradBucket.arm_early)
        head = [p for n, p in model.named_parameters() if n.startswith("head.") and p.requires_grad]
        self.bucket = D.GradBucket(params, early=head)
        self._has_bn = any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in model.modules())
        self._sim_scope = 0
        self.optim = torch.optim.Adam(params, lr=lr, betas=betas, eps=eps)  # trainer.py:141, optimizer.py:14-35

    def _train_mode(self):
        self.net.train()
        if not self.frozen_bn_batch_stats and isinstance(getattr(self.net, "upsampler", None), torch.nn.Module):
            self.net.upsampler.eval()

    def batch_forward(self, batch: Dict, num_iters=None):
        """trainer.py:377-477 (training branch)."""
        image, gt_mask, points = batch["images"], batch["instances"], batch["points"]
        prev_output = torch.zeros_like(image[:, :1], dtype=torch.float32)
        # the simulated-click forwards see one image: what the upsampler derives from the guidance alone (LoftUp's image
        # branch = 42 % of its FLOPs, FeatUp-JBU's records, LiFT's pyramid) is computed by the first and reused by the others
        self._sim_scope += 1
        with torch.no_grad(), guidance_scope(("sim_clicks", id(self), self._sim_scope)):
            if num_iters is None:
                num_iters = random.randint(0, self.max_num_next_clicks)
            for click_indx in range(num_iters):
                self.net.eval()
                net_input = torch.cat((image, prev_output), dim=1) if self.net.with_prev_mask else image
                prev_output = torch.sigmoid(self.net(net_input, points)["instances"])
                points = get_next_points(prev_output, gt_mask, points, click_indx + 1)
                self._train_mode()
            if self.net.with_prev_mask and self.prev_mask_drop_prob > 0 and num_iters > 0:
                zero = torch.from_numpy(np.random.random(size=prev_output.size(0)) < self.prev_mask_drop_prob)
                prev_output[zero.to(prev_output.device)] = 0
        # release what the scope memoised: the next step has another image, and the eval-mode packed weights (the cache's
        # geometry key) are rebuilt after every train forward -- kept slots would pile up (8 x 6 GB at configs[4])
        for m in self.net.modules():
            gc = getattr(m, "_gcache", None)
            if gc is not None:
                gc.clear()
        net_input = torch.cat((image, prev_output), dim=1) if self.net.with_prev_mask else image
        output = self.net(net_input, points)
        loss = self.loss_fn(output["instances"], gt_mask).mean()
        return loss, output

    def validate(self, batch: Dict, num_iters=None):
        """The validation branch of batch_forward (trainer.py:377-477 with validation=True, called from trainer.py:316-375):
        the net in eval mode throughout, no gradients, the same simulated corrective clicks; returns the loss."""
        self.net.eval()
        image, gt_mask, points = batch["images"], batch["instances"], batch["points"]
        prev_output = torch.zeros_like(image[:, :1], dtype=torch.float32)
        self._sim_scope += 1
        with torch.no_grad(), guidance_scope(("val_clicks", id(self), self._sim_scope)):
            if num_iters is None:
                num_iters = random.randint(0, self.max_num_next_clicks)
            for click_indx in range(num_iters):
                net_input = torch.cat((image, prev_output), dim=1) if self.net.with_prev_mask else image
                prev_output = torch.sigmoid(self.net(net_input, points)["instances"])
                points = get_next_points(prev_output, gt_mask, points, click_indx + 1)
            net_input = torch.cat((image, prev_output), dim=1) if self.net.with_prev_mask else image
            output = self.net(net_input, points)
            loss = self.loss_fn(output["instances"], gt_mask).mean()
        for m in self.net.modules():
            gc = getattr(m, "_gcache", None)
            if gc is not None:
                gc.clear()
        return loss.detach()

    def step(self, batch: Dict, num_iters=None):
        """One optimisation step; returns the (rank-local) loss as a 0-dim tensor."""
        self._train_mode()
        if self._has_bn and self.frozen_bn_batch_stats:
            # DDP(broadcast_buffers=True): rank 0's running statistics before every training forward, so the eval-mode
            # click-simulation forwards of all replicas normalise with the same numbers
            D.broadcast_buffers(self.net)
        self.bucket.zero()
        loss, _ = self.batch_forward(batch, num_iters)
        self.bucket.arm_early()
        loss.backward()
        self.bucket.check_bound()     # a zero_grad(set_to_none=True) by the caller would have detached the views
        self.bucket.finish_overlapped()  # head slice was issued inside backward; the rest + wait + /world here
        self.optim.step()
        return loss.detach()


def load_weights(model, path_to_weights: str):
    """core/training/trainer.py:621-626: the checkpoint's tensors laid over the model's own state dict (a probe checkpoint holds
    embed_coords.* and head.* only), strict=False."""
    current = model.state_dict()
    current.update(torch.load(path_to_weights, map_location="cpu", weights_only=False)["state_dict"])
    return model.load_state_dict(current, strict=False)


class EpochTrainer:
    """The epoch loop of the reference's iSegTrainer (core/training/trainer.py:180-314: ``run`` / ``training``) around a
    step object: for every epoch -- sampler.set_epoch, one optimisation step per batch of this rank's loader, the losses
    reduced to rank 0 for the log, ``last_checkpoint.pth`` after every epoch and ``<epoch:03d>.pth`` on the
    ``checkpoint_interval`` cadence ([[start_epoch, every], ...], train_cfg.yaml:23: the last entry whose start is <= the
    epoch applies), then the MultiStepLR(milestones, gamma 0.1) scheduler (models/defaults.py:110-114) steps once.

    ``stepper``: anything with ``.net`` (``get_state_dict_to_save`` / ``_config`` for the checkpoint writer), ``.optim`` and
    ``.step(batch) -> 0-dim loss tensor`` -- ``DataParallelTrainer`` on the GPU.  With ``val_loader`` every epoch ends with the reference's
    validation pass (trainer.py:316-375: eval mode, no gradients, mean loss).  Tensorboard / wandb logging, the AdaptiveIoU
    train metric and image dumps of the reference's loop are outside this path."""

    def __init__(self, stepper, loader, checkpoints_path=None, lr_milestones=(17, 20), checkpoint_interval=((0, 3), (15, 1)),
                 device=None, log=print, prefix="", val_loader=None):
        self.stepper, self.loader, self.checkpoints_path = stepper, loader, checkpoints_path
        self.val_loader = val_loader  # with it (and a stepper that has .validate): trainer.py:316-375 after every epoch
        self.val_history = []
        self.checkpoint_interval = [tuple(x) for x in checkpoint_interval] if isinstance(checkpoint_interval, (list, tuple)) \
            else checkpoint_interval
        self.device, self.log, self.prefix = device, log, prefix
        self.lr_scheduler = torch.optim.lr_scheduler.MultiStepLR(stepper.optim, milestones=list(lr_milestones), gamma=0.1)
        self.history = []  # (epoch, mean loss over the epoch on rank 0, learning rate)

    def run(self, num_epochs, start_epoch=0):
        for _ in range(start_epoch):  # resuming: the schedule continues where it stopped (train_cfg.yaml:33)
            self.lr_scheduler.step()
        for epoch in range(start_epoch, num_epochs):
            self.training(epoch)
            if self.val_loader is not None and hasattr(self.stepper, "validate"):
                self.validation(epoch)
        return self.history

    def validation(self, epoch):
        """trainer.py:316-375: mean validation loss over this rank's shard of the validation loader, reduced to rank 0."""
        total, n = 0.0, 0
        for batch in self.val_loader:
            if self.device is not None:
                batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
            red = D.reduce_loss_dict({"overall": self.stepper.validate(batch).float()})
            total, n = total + float(red["overall"]), n + 1
        if D.get_rank() == 0:
            self.val_history.append((epoch, total / max(n, 1)))
            self.log(f"Epoch {epoch}, validation loss: {total / max(n, 1):.4f}")

    def _interval(self, epoch):
        ci = self.checkpoint_interval
        return [x for x in ci if x[0] <= epoch][-1][1] if isinstance(ci, (list, tuple)) else ci

    def training(self, epoch):
        from ..utils.misc import save_checkpoint
        sampler = getattr(self.loader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)
        total, n = 0.0, 0
        for batch in self.loader:
            if self.device is not None:
                batch = {k: v.to(self.device, non_blocking=True) for k, v in batch.items()}
            loss = self.stepper.step(batch)
            red = D.reduce_loss_dict({"overall": loss.detach().float()})
            total, n = total + float(red["overall"]), n + 1
        lr = self.stepper.optim.param_groups[0]["lr"]
        if D.get_rank() == 0:
            self.history.append((epoch, total / max(n, 1), lr))
            self.log(f"Epoch {epoch}, training loss {total / max(n, 1):.4f}, lr {lr:.2e}, {n} steps")
            if self.checkpoints_path is not None:
                save_checkpoint(self.stepper.net, self.checkpoints_path, prefix=self.prefix, epoch=None, verbose=False)
                if epoch % self._interval(epoch) == 0:
                    save_checkpoint(self.stepper.net, self.checkpoints_path, prefix=self.prefix, epoch=epoch, verbose=False)
        self.lr_scheduler.step()
