"""Data-parallel helpers (reference core/utils/distributed.py:8-78, core/utils/exp.py:33-36).

One process per GPU over ``torch.distributed`` (backend "nccl" IS RCCL on ROCm; "gloo" for the
CPU tests).  The reference wraps the net in DistributedDataParallel, whose bucketed all-reduce
fires inside ``loss.backward()``; only ``embed_coords`` and ``head`` are trainable there (2.88 M
fp32 = 11.5 MB for DINOv2-S/14), so here the gradients of the trainable parameters are flattened
into ONE bucket and all-reduced once per step (``GradBucket``) -- on xGMI a ring all-reduce of
11.5 MB is ~0.13 ms, far below a step, so no finer bucketing or tree algorithm is warranted.
The rank's device comes from ``LOCAL_RANK`` (the reference reads it from YAML only)."""
import os

import torch
from torch import distributed as dist
from torch.utils import data


# ISEGPROBE_GRAD_OVERLAP=0: one all-reduce of the whole bucket after backward instead of the head slice from inside backward
GRAD_OVERLAP = os.environ.get("ISEGPROBE_GRAD_OVERLAP", "1") != "0"


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def init_distributed(backend=None):
    """init_process_group from the torchrun environment (no-op for a single process)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world < 2 or (dist.is_available() and dist.is_initialized()):
        return world > 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(get_local_rank())
    dist.init_process_group(backend=backend, init_method="env://")
    return True


def synchronize():
    if get_world_size() > 1:
        dist.barrier()


def reduce_loss_dict(loss_dict):
    """Mean of scalar losses on rank 0 (distributed.py:31-53): one dist.reduce of the stacked scalars."""
    world_size = get_world_size()
    if world_size < 2:
        return loss_dict
    with torch.no_grad():
        keys = list(loss_dict.keys())
        losses = torch.stack([loss_dict[k] for k in keys], 0)
        dist.reduce(losses, dst=0)
        if dist.get_rank() == 0:
            losses /= world_size
        return {k: v for k, v in zip(keys, losses)}


def get_sampler(dataset, shuffle, distributed, generator=None, seed=0):
    if distributed:
        return data.distributed.DistributedSampler(dataset, shuffle=shuffle, seed=seed)
    return data.RandomSampler(dataset, generator=generator) if shuffle else data.SequentialSampler(dataset)


def shard_indices(n, rank=None, world=None):
    """Disjoint contiguous-stride shard of range(n) for this rank (DistributedSampler's layout without
    shuffling: rank, rank+world, ...; the tail is padded by wrapping so every rank has equal work)."""
    rank = get_rank() if rank is None else rank
    world = get_world_size() if world is None else world
    per = (n + world - 1) // world
    idx = list(range(rank, per * world, world))
    return [i % n for i in idx]


def broadcast_buffers(module, src=0):
    """DDP's ``broadcast_buffers=True`` (the reference wraps the net with the default, core/utils/distributed.py:66-78):
    before every training forward rank ``src``'s module buffers replace every other rank's -- here the running
    statistics of the frozen LiFT / LoftUp BatchNorms, which train-mode forwards update from each rank's own shard, and
    their integer ``num_batches_tracked`` counters.  Two flat broadcasts (floating point, integer), no device->host
    synchronisation: the receiving ranks copy unconditionally (the buffers' version counters move, as they do anyway
    in the train-mode forward that follows)."""
    if get_world_size() < 2:
        return 0
    done = 0
    for floating in (True, False):
        bufs = [b for b in module.buffers() if b.numel() > 0 and b.is_floating_point() == floating and not b.is_complex()]
        if not bufs:
            continue
        flat = torch.cat([b.detach().reshape(-1).to(torch.float32 if floating else torch.int64) for b in bufs])
        dist.broadcast(flat, src=src)
        if get_rank() != src:
            off = 0
            with torch.no_grad():
                for b in bufs:
                    n = b.numel()
                    b.copy_(flat[off:off + n].view_as(b))
                    off += n
        done += len(bufs)
    return done


class GradBucket:
    """Flat bucket of the trainable parameters' gradients: one all-reduce(sum) per step, then /world.

    ``params`` order is fixed at construction (must match on every rank).  The bucket is a single
    contiguous tensor on the parameters' device; ``.grad`` of each parameter becomes a view into it, so
    backward writes straight into the bucket and no copy precedes the collective.

    Do NOT call ``optimizer.zero_grad()`` / ``model.zero_grad()`` on these parameters (torch's default
    ``set_to_none=True`` drops the views and the all-reduce would then average stale zeros): use ``zero()``, which
    also re-binds any view that was dropped; ``check_bound()`` raises if a gradient no longer aliases the bucket."""

    def __init__(self, params, dtype=torch.float32, early=()):
        """``early``: the parameters whose gradients are final first in backward (the seg head: its weight gradients are
        complete before the upsampler's / trunk's data gradient starts).  They are laid out at the front of the bucket
        and their slice is all-reduced asynchronously the moment the last of them has been accumulated
        (``arm_early`` / ``finish``), overlapping the collective with the rest of backward -- what the reference's DDP
        does with its bucketed hooks inside ``loss.backward()`` (core/utils/distributed.py:66-78)."""
        early_ids = {id(p) for p in early if p.requires_grad}
        params = [p for p in params if p.requires_grad]
        self.params = [p for p in params if id(p) in early_ids] + [p for p in params if id(p) not in early_ids]
        self.n_early = sum(1 for p in params if id(p) in early_ids)
        self._early_numel = sum(p.numel() for p in self.params[:self.n_early])
        self._pending = None   # (remaining early params, work handle) of the step in flight
        self._hooks = []
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(sizes), device=dev, dtype=dtype)
        self._offsets = []
        off = 0
        for p, n in zip(self.params, sizes):
            self._offsets.append(off)
            off += n
        self._bind()

    def _bind(self):
        for p, off in zip(self.params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                p.grad = self.flat[off:off + p.numel()].view_as(p)

    def check_bound(self):
        for p, off in zip(self.params, self._offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
                raise RuntimeError("GradBucket: a parameter's .grad no longer aliases the flat bucket (zero_grad(set_to_none=True)"
                                   " or a re-assigned .grad); use GradBucket.zero() instead of zero_grad()")

    def nbytes(self):
        return self.flat.numel() * self.flat.element_size()

    def zero(self):
        self.flat.zero_()
        self._bind()

    def all_reduce_mean(self, async_op=False):
        """Average gradients across ranks (what DDP does); returns the work handle when async."""
        world = get_world_size()
        if world < 2:
            return None
        work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
        if async_op:
            return work
        self.flat.div_(world)
        return None

    def finish(self, work):
        if work is not None:
            work.wait()
            self.flat.div_(get_world_size())

    # ---- overlap of the early slice with the rest of backward
    def arm_early(self):
        """Call before ``loss.backward()``: when the last early parameter's gradient has been accumulated, the early
        slice's all-reduce(sum) is issued with ``async_op=True`` (on RCCL it runs on the communicator's stream behind
        an event of the compute stream).  No-op for a single process or without early parameters."""
        if get_world_size() < 2 or not self.n_early or not GRAD_OVERLAP:
            self._pending = None
            return
        if not self._hooks:
            for p in self.params[:self.n_early]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_early_grad))
        self._pending = [self.n_early, None]

    def _on_early_grad(self, param):
        if self._pending is None:
            return
        self._pending[0] -= 1
        if self._pending[0] == 0:
            self._pending[1] = dist.all_reduce(self.flat[:self._early_numel], op=dist.ReduceOp.SUM, async_op=True)

    def finish_overlapped(self):
        """After backward: all-reduce what the early launch did not cover, wait for both, divide by the world size.
        Equals ``all_reduce_mean()`` bit for bit on two ranks (the sum of two floats is order-independent) and up to
        the collective's own reduction order beyond.

        Every rank issues the SAME sequence of collectives whatever its autograd graph did: [early slice, rest].  A rank
        whose early hook did not fire the expected number of times (a head parameter that received no gradient this
        step) issues the early slice here, synchronously, instead of skipping it -- a skipped collective would pair
        this rank's next all-reduce with the other ranks' early one and hang the job."""
        world = get_world_size()
        if world < 2:
            return
        pending, self._pending = self._pending, None
        if pending is None:  # arm_early() was not called (or there are no early parameters): one collective
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(world)
            return
        early_work = pending[1]
        self.early_fired_in_backward = early_work is not None
        if early_work is None:
            dist.all_reduce(self.flat[:self._early_numel], op=dist.ReduceOp.SUM)
        if self._early_numel < self.flat.numel():
            dist.all_reduce(self.flat[self._early_numel:], op=dist.ReduceOp.SUM)
        if early_work is not None:
            early_work.wait()
        self.flat.div_(world)
