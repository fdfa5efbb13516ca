#!/usr/bin/env python3
"""Training entry point for the probe on the HIP path (the reference's train.py:13-27 drives a
Hydra config + a model script; this one keeps the same roles with plain arguments).

    python train.py --steps 20                                   # single GPU
    python -m torch.distributed.run --nproc-per-node 8 train.py  # data parallel, RCCL over xGMI
    python train.py +exp.name=my_name +exp.model_path=models/sbd/dinov2/patch-embed_loftup.py dataloader.batch_size=8
                                                                 # the reference's Hydra form (README.md:85-90), parsed
                                                                 # without hydra (core/utils/overrides.py)

`--model` names one of the reference's model scripts (models/sbd/<family>/<script>.py: backbone, click-injection
mode, click encoder, upsampler and head exactly as configured there).  Each rank builds the same model (frozen
backbone + frozen upsampler, trainable embed_coords + head), draws its own shard of the (synthetic, SBD-shaped) minibatch, and takes optimisation steps
with ONE flat-bucket gradient all-reduce per step (core/training/trainer.py).

    python train.py --dataset /data/SBD/dataset --epochs 20 --save ckpts/    # the reference's loop on an SBD tree
    python train.py +exp.model_path=models/sbd/dinov2/patch-embed_loftup.py +datasets.SBD_PATH=/data/SBD/dataset

With --dataset the step runs inside the reference's epoch loop (core/training/trainer.py::EpochTrainer: SBD train reader,
MultiPointSampler clicks, per-rank shards, MultiStepLR milestones, last_checkpoint.pth + NNN.pth cadence; core/data/)."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def synthetic_batch(B, S, rng, P=24, device="cuda"):
    yy, xx = np.mgrid[:S, :S]
    images = torch.rand(B, 3, S, S)
    gts, pts = [], -np.ones((B, 2 * P, 3), np.float32)
    for b in range(B):
        cy, cx, ry, rx = rng.uniform(0.3, 0.7) * S, rng.uniform(0.3, 0.7) * S, rng.uniform(0.1, 0.3) * S, rng.uniform(0.1, 0.3) * S
        m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.float32)
        images[b] += torch.from_numpy(m)[None] * 0.5
        gts.append(torch.from_numpy(m)[None])
        pts[b, 0] = (int(cy), int(cx), 0)  # one positive click at the object centre
    return {"images": images.clamp(0, 1).to(device), "instances": torch.stack(gts).to(device),
            "points": torch.from_numpy(pts).to(device)}


def model_configs(name, size, arch="dinov2_vits14", upsampler=None, injection=None):
    """backbone / embed_coords / head / upsampler configs of the reference's model scripts
    (models/sbd/{dinov2,vit,maskclip}/*.py: define_modules_cfg)."""
    S = (size, size)
    dim = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024}[arch]
    patch_embed = lambda p, d: {"type": "patchEmbed", "params": dict(img_size=S, patch_size=(p, p), embed_dim=d)}
    head = lambda c: {"type": "convhead", "params": dict(in_channels=c, num_layers=2, num_classes=1)}
    dinov2 = lambda inj: {"type": "dinov2", "params": {"arch": arch, "feats_injection_mode": injection or inj}}
    ups = {"bilinear": {"type": "bilinear", "params": None}, "noup": {"type": "identity", "params": None},
           "jbu": {"type": "jbu_featup", "params": dict(backbone_type="dinov2", use_norm=True, feat_dim=dim)},
           "lift": {"type": "lift", "params": dict(lift_path=None, n_dim=dim, patch=14)},
           "loftup": {"type": "loftup", "params": dict(upsampler_path=None, n_dim=dim, lr_pe_type="sine", lr_size=16)}}
    family, _, script = name.replace("models/sbd/", "").replace(".py", "").partition("/")
    if family == "dinov2" and script.startswith("patch-embed_"):
        cfg = dict(backbone_cfg=dinov2("before_backbone"), embed_coords_cfg=patch_embed(14, dim), head_cfg=head(dim),
                   upsampler_cfg=ups[script.split("_", 1)[1]])
    elif family == "dinov2" and script == "simple-vit_noup":
        cfg = dict(backbone_cfg=dinov2("after_backbone"), head_cfg=head(dim), upsampler_cfg=ups["noup"],
                   embed_coords_cfg={"type": "simple_vit", "params": dict(img_size=list(S), patch_size=(14, 14), embed_dim=dim,
                                                                           depth=6, heads=8, mlp_dim=2048, channels=3, dim_head=64)})
    elif family == "vit" and script == "patch-embed_noup":
        cfg = dict(backbone_cfg={"type": "vit", "params": dict(arch="vit_small_patch16_224", patch_size=16, feat_type="key",
                                                               feats_injection_mode=injection or "before_backbone")},
                   embed_coords_cfg=patch_embed(16, 384), head_cfg=head(384), upsampler_cfg=ups["noup"])
    elif family == "maskclip" and script == "patch-embed_noup":
        cfg = dict(backbone_cfg={"type": "mask_clip", "params": dict(model_name="ViT-B/16",
                                                                     feats_injection_mode=injection or "before_backbone")},
                   embed_coords_cfg=patch_embed(16, 768), head_cfg=head(512), upsampler_cfg=ups["noup"])
    else:
        raise SystemExit(f"unknown model script {name!r}")
    if upsampler is not None:
        cfg["upsampler_cfg"] = ups.get(upsampler, {"type": upsampler, "params": None})
    return cfg


def D_rank0():
    return int(os.environ.get("RANK", "0")) == 0


def find_resume_checkpoint(exp_dir, prefix):
    """The checkpoint a resumed experiment continues from (reference trainer.py:559-568): exactly one file
    ``<exp_dir>/checkpoints/<prefix>*.pth`` -- the layout ``init_experiment`` gives an experiment (exp.py:53-56) -- or, for a
    bare checkpoint directory such as --save writes, ``<exp_dir>/<prefix>*.pth``.  None or several matches are an error, as there."""
    from pathlib import Path
    root = Path(exp_dir)
    where = root / "checkpoints" if (root / "checkpoints").is_dir() else root
    found = sorted(where.glob(f"{prefix}*.pth"))
    if len(found) != 1:
        raise SystemExit(f"resume: {len(found)} checkpoints match {where}/{prefix}*.pth (exactly one expected)"
                         + "".join(f"\n  {f}" for f in found))
    return str(found[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (reference train_cfg.yaml:18)")
    ap.add_argument("--size", type=int, default=224, help="crop size (reference train_cfg.yaml:22)")
    ap.add_argument("--arch", default="dinov2_vits14")
    ap.add_argument("--model", default="dinov2/patch-embed_bilinear",
                    help="reference model script: dinov2/patch-embed_{bilinear,jbu,lift,loftup,noup}, dinov2/simple-vit_noup, "
                         "vit/patch-embed_noup, maskclip/patch-embed_noup (reference: +exp.model_path=models/sbd/<this>.py)")
    ap.add_argument("--upsampler", default=None, help="override the script's upsampler")
    ap.add_argument("--injection", default=None, help="override feats_injection_mode (before_backbone | after_backbone)")
    ap.add_argument("--lr", type=float, default=5e-5)
    ap.add_argument("--save", default=None, help="directory for a reference-format last_checkpoint.pth (rank 0)")
    ap.add_argument("--dataset", default=None,
                    help="SBD root (img/, inst/, train.txt): train on it for --epochs epochs instead of synthetic batches "
                         "(reference: DATASETS.SBD_PATH of configs/main_cfg.yaml; Hydra form +datasets.SBD_PATH=<root>)")
    ap.add_argument("--main-cfg", default=None, help="the reference's configs/main_cfg.yaml: --dataset defaults to its DATASETS.SBD_PATH "
                                                     "(models/defaults.py:82), the sampling-weights pickle to ./assets/sbd_samples_weights.pkl if present")
    ap.add_argument("--epochs", type=int, default=None, help="with --dataset: training_params.epochs (train_cfg.yaml:21)")
    ap.add_argument("--epoch-len", type=int, default=-1, help="with --dataset: samples per epoch (-1: the dataset's size)")
    ap.add_argument("--workers", type=int, default=None, help="with --dataset: DataLoader workers (dataloader.workers)")
    ap.add_argument("--weights", default=None, help="checkpoint whose tensors initialise the model (reference training.weights, trainer.py:550-557)")
    ap.add_argument("--resume-exp", default=None, help="experiment directory to continue (reference training.resume_exp): loads the ONE "
                                                       "checkpoint <dir>/checkpoints/<--resume-prefix>*.pth (trainer.py:559-568); --weights wins")
    ap.add_argument("--resume-prefix", default="latest", help="reference training.resume_prefix (train_cfg.yaml:33), e.g. last_checkpoint or 004")
    ap.add_argument("--start-epoch", type=int, default=0, help="with --dataset: continue at this epoch (training.start_epoch: the LR "
                                                               "schedule is advanced to it, trainer.py:168-170)")
    ap.add_argument("--validate", action="store_true", help="with --dataset: a validation pass over <root>/val.txt after every epoch "
                                                            "(reference training_params.do_validation)")
    ap.add_argument("--val-len", type=int, default=-1, help="with --validate: samples per validation pass (-1: the split's size)")
    ap.add_argument("--samples-scores", default=None,
                    help="with --dataset: the sampling-weights pickle (reference ./assets/sbd_samples_weights.pkl, gamma 1.25)")
    ap.add_argument("--eval-frozen-bn", action="store_true",
                    help="keep the frozen upsampler's BatchNorm in eval mode (the reference's net.train() uses batch statistics)")
    from isegprobe_amd.core.utils.overrides import TRAIN_DEFAULTS, apply_overrides, split_overrides
    overrides, rest = split_overrides(sys.argv[1:])
    args = ap.parse_args(rest)
    if overrides:  # Hydra-style tokens over configs/train_cfg.yaml's keys
        cfg = apply_overrides(TRAIN_DEFAULTS, overrides)
        given = {k for k, _, _ in overrides}
        if "exp.model_path" in given:
            args.model = str(cfg["exp"]["model_path"])
        if "dataloader.batch_size" in given:  # global batch, split over the GPUs (trainer.py:67-68)
            args.batch = max(1, int(cfg["dataloader"]["batch_size"]) // int(os.environ.get("WORLD_SIZE", "1")))
        if "training_params.crop_size" in given:
            cs = cfg["training_params"]["crop_size"]
            args.size = int(cs[0] if isinstance(cs, (list, tuple)) else cs)
        if "training.local_rank" in given:  # the reference reads the rank's device from YAML only (train_cfg.yaml:36)
            os.environ.setdefault("LOCAL_RANK", str(cfg["training"]["local_rank"]))
        if cfg["training"].get("weights"):
            args.weights = str(cfg["training"]["weights"])
        if "training.start_epoch" in given:
            args.start_epoch = int(cfg["training"]["start_epoch"])
        if cfg["training"].get("resume_exp"):
            args.resume_exp = str(cfg["training"]["resume_exp"])
        if "training.resume_prefix" in given:
            args.resume_prefix = str(cfg["training"]["resume_prefix"])
        if "datasets.SBD_PATH" in given:
            args.dataset = str(cfg["datasets"]["SBD_PATH"])
        tp = cfg["training_params"]
        args.epochs = args.epochs if args.epochs is not None else int(tp["epochs"])
        args.workers = args.workers if args.workers is not None else int(cfg["dataloader"]["workers"])
        args.lr_milestones, args.checkpoint_interval = list(tp["lr_milestones"]), [tuple(x) for x in tp["checkpoint_interval"]]
        args.num_max_points, args.seed = int(tp["num_max_points"]), int(cfg["training"]["seed"])
        if D_rank0():
            print(f"experiment '{cfg['exp']['name']}', model script {args.model}")
    for k, v in (("lr_milestones", TRAIN_DEFAULTS["training_params"]["lr_milestones"]), ("num_max_points", 24), ("seed", 0),
                 ("checkpoint_interval", [tuple(x) for x in TRAIN_DEFAULTS["training_params"]["checkpoint_interval"]])):
        if not hasattr(args, k):
            setattr(args, k, v)
    args.epochs = args.epochs if args.epochs is not None else TRAIN_DEFAULTS["training_params"]["epochs"]
    args.workers = args.workers if args.workers is not None else TRAIN_DEFAULTS["dataloader"]["workers"]

    if args.main_cfg and not args.dataset:
        import yaml
        args.dataset = str((yaml.safe_load(open(args.main_cfg)) or {}).get("DATASETS", {}).get("SBD_PATH") or "") or None
        if args.dataset is None:
            raise SystemExit(f"{args.main_cfg}: no DATASETS.SBD_PATH")
        if args.samples_scores is None and os.path.exists("./assets/sbd_samples_weights.pkl"):
            args.samples_scores = "./assets/sbd_samples_weights.pkl"  # models/defaults.py:88
    from isegprobe_amd.core.model import iSegProbeModel
    from isegprobe_amd.core.training.trainer import DataParallelTrainer
    from isegprobe_amd.core.utils import distributed as D

    # ISEGPROBE_DIST_BACKEND=gloo ISEGPROBE_SHARE_GPU=1: rehearsal of the multi-rank run on a one-GPU box (every rank on device 0,
    # gloo over device tensors in place of RCCL, which wants one device per rank) -- as bench.py and evaluate.py take them
    share = os.environ.get("ISEGPROBE_SHARE_GPU", "0") == "1"
    if share:
        torch.cuda.set_device(0)
    distributed = D.init_distributed(os.environ.get("ISEGPROBE_DIST_BACKEND") or None)
    torch.cuda.set_device(0 if share else D.get_local_rank())
    torch.manual_seed(0)  # identical initial weights on every rank
    model = iSegProbeModel(**model_configs(args.model, args.size, args.arch, args.upsampler, args.injection),
                           use_disks=True, norm_radius=5, with_prev_mask=True).cuda()
    if args.weights is None and args.resume_exp:
        args.weights = find_resume_checkpoint(args.resume_exp, args.resume_prefix)
    if args.weights:
        from isegprobe_amd.core.training.trainer import load_weights
        msg = load_weights(model, args.weights)
        if D_rank0():
            print(f"Loaded weights from {args.weights} with msg: {msg}")
    trainer = DataParallelTrainer(model, lr=args.lr, frozen_bn_batch_stats=not args.eval_frozen_bn)
    if args.dataset:
        # the reference's loop (trainer.py:180-314): epochs over the SBD train split, clicks from MultiPointSampler, this
        # rank's shard of every epoch, LR milestones and checkpoint cadence of train_cfg.yaml
        import random
        from isegprobe_amd.core.data import SBDTrainSet, make_loader
        from isegprobe_amd.core.training.trainer import EpochTrainer
        if args.seed >= 0:
            random.seed(args.seed + D.get_rank()), np.random.seed(args.seed + D.get_rank())
        trainset = SBDTrainSet(args.dataset, crop_size=(args.size, args.size), num_max_points=args.num_max_points,
                               samples_scores_path=args.samples_scores, epoch_len=args.epoch_len)
        loader = make_loader(trainset, args.batch, workers=args.workers, seed=max(args.seed, 0))
        model.save_cfg = {"embed_coords": True, "backbone": False, "upsampler": False, "head": True}  # as the model scripts do
        if D.get_rank() == 0:
            print(f"model {args.model}  world {D.get_world_size()}  SBD train: {len(trainset)} samples per epoch, "
                  f"{len(loader)} steps per rank and epoch at batch {args.batch}, {args.epochs} epochs")
        val_loader = None
        if args.validate:  # training_params.do_validation (train_cfg.yaml:25): the val split through the same feed
            valset = SBDTrainSet(args.dataset, crop_size=(args.size, args.size), num_max_points=args.num_max_points, split="val",
                                 epoch_len=args.val_len)
            val_loader = make_loader(valset, args.batch, workers=args.workers, seed=max(args.seed, 0), shuffle=False)
        EpochTrainer(trainer, loader, checkpoints_path=args.save, lr_milestones=args.lr_milestones,
                     checkpoint_interval=args.checkpoint_interval, device="cuda", val_loader=val_loader).run(args.epochs, start_epoch=args.start_epoch)
        D.synchronize()
        return
    rng = np.random.default_rng(100 + D.get_rank())
    if D.get_rank() == 0:
        print(f"model {args.model}  world {D.get_world_size()}  trainable bucket {trainer.bucket.nbytes() / 1e6:.1f} MB  "
              f"per-GPU batch {args.batch} @ {args.size}^2")
    for step in range(args.steps):
        batch = synthetic_batch(args.batch, args.size, rng)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = trainer.step(batch)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        red = D.reduce_loss_dict({"loss": loss})
        if D.get_rank() == 0:
            print(f"step {step:3d}  loss {float(red['loss']):.4f}  {dt * 1e3:7.1f} ms  "
                  f"{D.get_world_size() * args.batch / dt:7.1f} img/s")
    if args.save and D.get_rank() == 0:
        from isegprobe_amd.core.utils.misc import save_checkpoint
        model.save_cfg = {"embed_coords": True, "backbone": False, "upsampler": False, "head": True}  # as the model scripts do
        save_checkpoint(model, args.save)
    D.synchronize()


if __name__ == "__main__":
    main()
