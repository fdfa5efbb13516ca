#!/usr/bin/env python3
"""NoC evaluation entry point on the HIP path (the reference's evaluate.py:30-221 is a Hydra script
around the same loop: load model -> get_predictor(NoBRS, flip, zoom-in fixed<S>) -> evaluate_dataset ->
NoC table).

    python evaluate.py --dataset /path/to/GrabCut --checkpoint ckpt.pth --eval-mode fixed224
    python evaluate.py --synthetic 50                 # GrabCut-layout fixture of seeded ellipses
    python evaluate.py +checkpoint=/path/to/ckpt +datasets=GrabCut,Berkeley eval_mode=fixed224 n_clicks=20
                                                      # the reference's Hydra form (README.md:97-103; keys of
                                                      # configs/eval_cfg.yaml; dataset roots from main_cfg_path's DATASETS)
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 evaluate.py --dataset ... --logs-path ...
                                                      # images sharded over the GPUs of the node (one process per GPU, no
                                                      # data-path collective); rank 0 gathers the IoU arrays and writes the
                                                      # same table a single process would
"""
import argparse
import os
import sys
import tempfile
from datetime import timedelta

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main(argv=None):
    """Returns [(dataset name, per-object IoU arrays, results dict of the printed table row)] (tests call it in-process)."""
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default=None, help="dataset root directory")
    ap.add_argument("--dataset-name", default="GrabCut",
                    help="reader: GrabCut | Berkeley | DAVIS | COCO_MVal | SBD | SBD_Train | PascalVOC (inference/utils.py:86-104)")
    ap.add_argument("--synthetic", type=int, default=0, help="evaluate on N synthetic GrabCut-layout samples")
    ap.add_argument("--checkpoint", default=None, help="reference-format checkpoint {'state_dict','config'}")
    ap.add_argument("--arch", default="dinov2_vits14")
    ap.add_argument("--upsampler", default="bilinear")
    ap.add_argument("--eval-mode", default="fixed224",
                    help="fixed<H>[,<W>] or cvpr (448x448, DAVIS 672x672; reference eval_cfg.yaml:36, inference/utils.py:301-316)")
    ap.add_argument("--n-clicks", type=int, default=20)
    ap.add_argument("--thresh", type=float, default=0.5)
    ap.add_argument("--target-iou", type=float, default=0.90)
    ap.add_argument("--clicks-limit", type=int, default=None,
                    help="feed the network at most this many clicks of each polarity (-1 = n_clicks; eval_cfg.yaml clicks_limit, "
                         "inference/utils.py:286-289 -> predictor net_clicks_limit)")
    ap.add_argument("--min-n-clicks", type=int, default=1, help="clicks before the IoU target may stop an object (eval_cfg.yaml min_n_clicks)")
    ap.add_argument("--logs", "--logs-path", dest="logs", default=None,
                    help="directory for the results table / IoU pickles (default: a temp dir).  Under torchrun write --logs-path: "
                         "its own parser claims `--logs` as an abbreviation of --logs-specs")
    ap.add_argument("--host-clicker", action="store_true",
                    help="robot user + IoU on the host (numpy/scipy) as in the reference, instead of the device clicker")
    ap.add_argument("--save-feats", type=int, default=0, metavar="N",
                    help="dump the low- / high-resolution features of the first click of the first N images under "
                         "<logs>/feats/<dataset>/ (eval_cfg.yaml save_feats / save_feats_for_n_imgs, inference/utils.py:587-627)")
    ap.add_argument("--fp32", action="store_true",
                    help="checking mode: fp32-accurate arithmetic (model.forward_fp32, three bf16 products; 3-4x slower)")
    from isegprobe_amd.core.utils.overrides import DATASET_PATH_KEYS, EVAL_DEFAULTS, apply_overrides, split_overrides
    overrides, rest = split_overrides(argv)
    args = ap.parse_args(rest)
    jobs = None  # [(dataset name, root)]
    feats_folder = "features"
    print_ious, iou_analysis = True, False
    if overrides:
        cfg = apply_overrides(EVAL_DEFAULTS, overrides)
        given = {k for k, _, _ in overrides}
        if cfg["mode"] != "NoBRS":
            raise SystemExit("only mode=NoBRS is built (every experiment of the reference used it, eval_cfg.yaml:13-14)")
        if cfg["eval_ritm"]:
            raise SystemExit("eval_ritm=true is outside the probed path")
        args.checkpoint = cfg["checkpoint"] if "checkpoint" in given else args.checkpoint
        args.eval_mode, args.n_clicks, args.thresh = str(cfg["eval_mode"]), int(cfg["n_clicks"]), float(cfg["thresh"])
        args.target_iou, print_ious = float(cfg["target_iou"]), bool(cfg["print_ious"])
        args.clicks_limit = None if cfg["clicks_limit"] is None else int(cfg["clicks_limit"])
        args.min_n_clicks, iou_analysis = int(cfg["min_n_clicks"]), bool(cfg["iou_analysis"])
        if cfg["logs_path"]:
            args.logs = str(cfg["logs_path"])
        if cfg["save_feats"]:
            args.save_feats, feats_folder = int(cfg["save_feats_for_n_imgs"]), str(cfg["save_feats_folder_name"])
        if "datasets" in given and not args.dataset and not args.synthetic:
            import yaml
            if not os.path.exists(str(cfg["main_cfg_path"])):
                raise SystemExit(f"datasets={cfg['datasets']}: dataset roots come from {cfg['main_cfg_path']} (DATASETS.<NAME>_PATH), "
                                 "which does not exist; give --dataset ROOT --dataset-name NAME instead")
            roots = (yaml.safe_load(open(str(cfg["main_cfg_path"]))) or {}).get("DATASETS", {})
            jobs = [(n, roots[DATASET_PATH_KEYS[n]]) for n in str(cfg["datasets"]).split(",")]
    # inference/utils.py:254-257: printing the per-click IoUs forces every click to run; otherwise stop at target_iou >= 0.8
    max_iou_thr = 1.01 if ((iou_analysis or print_ious) and args.min_n_clicks <= 1) else max(0.8, args.target_iou)

    import isegprobe_amd
    from isegprobe_amd.core.inference.datasets import get_dataset, write_synthetic_grabcut
    from isegprobe_amd.core.inference.evaluation import evaluate_dataset
    from isegprobe_amd.core.inference.predictors import get_predictor
    from isegprobe_amd.core.inference.utils import compute_noc_metric
    from isegprobe_amd.core.model import iSegProbeModel
    from isegprobe_amd.core.utils.serialization import load_model

    from isegprobe_amd.core.inference.utils import get_zoom_in_params
    crop = get_zoom_in_params(args.eval_mode, args.dataset_name)["target_size"]  # (model construction: the first dataset's size)
    # Under torchrun every rank evaluates images rank, rank + world, ... on its own GPU.  The only exchange is the
    # gather of the per-object IoU arrays -- host objects, so the process group is gloo (RCCL would stage them through device
    # tensors for nothing).  ISEGPROBE_SHARE_GPU=1 puts every rank on device 0 (rehearsal on a one-GPU box).
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    shard = None
    if world > 1:
        from isegprobe_amd.core.utils import distributed as D
        D.init_distributed("gloo")
        shard = (rank, world)
        torch.cuda.set_device(0 if os.environ.get("ISEGPROBE_SHARE_GPU", "0") == "1" else D.get_local_rank())
    device = torch.device("cuda", torch.cuda.current_device())
    if args.checkpoint:
        from isegprobe_amd.core.inference.utils import load_is_model
        model = load_is_model(args.checkpoint, device)  # reference-format {"state_dict", "config"} (inference/utils.py:37-83)
    else:
        dim = {"dinov2_vits14": 384, "dinov2_vitb14": 768, "dinov2_vitl14": 1024}[args.arch]
        up_params = {"backbone_type": "dinov2"} if args.upsampler == "jbu_featup" else None
        torch.manual_seed(0)
        model = iSegProbeModel(
            backbone_cfg={"type": "dinov2", "params": {"arch": args.arch, "feats_injection_mode": "before_backbone"}},
            head_cfg={"type": "convhead", "params": dict(in_channels=dim, num_layers=2, num_classes=1)},
            embed_coords_cfg={"type": "patchEmbed", "params": dict(img_size=crop, patch_size=(14, 14), embed_dim=dim)},
            upsampler_cfg={"type": args.upsampler, "params": up_params},
            use_disks=True, norm_radius=5, with_prev_mask=True)
    model = model.to(device).eval()
    if args.fp32:  # the predictor calls net(image, points): route it through the fp32-accurate forward
        model.forward = model.forward_fp32

    tmp = None
    if args.synthetic:
        if world > 1:  # one tree for all ranks: rank 0 writes it, the others learn its path
            from torch import distributed as dist
            if rank == 0:
                tmp = tempfile.TemporaryDirectory()
                write_synthetic_grabcut(tmp.name, args.synthetic)
            box = [tmp.name if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            args.dataset = box[0]
        else:
            tmp = tempfile.TemporaryDirectory()
            args.dataset = str(write_synthetic_grabcut(tmp.name, args.synthetic))
    if jobs is None:
        if not args.dataset:
            raise SystemExit("give --dataset, --synthetic N or +datasets=...")
        jobs = [(args.dataset_name, args.dataset)]
    from isegprobe_amd.core.inference.utils import save_iou_analysis_data, save_results
    logs = (args.logs or tempfile.mkdtemp(prefix="isegprobe_eval_")) if rank == 0 else None
    out = []
    for i, (name, root) in enumerate(jobs):
        dataset = get_dataset(name, root)
        # evaluate.py:72-94: zoom-in parameters and the predictor are rebuilt per dataset (cvpr: DAVIS runs at 672 x 672)
        predictor_params = {}
        if args.clicks_limit is not None:  # inference/utils.py:286-289
            predictor_params["net_clicks_limit"] = args.n_clicks if args.clicks_limit == -1 else args.clicks_limit
        predictor = get_predictor(model, "NoBRS", device, prob_thresh=args.thresh, predictor_params=predictor_params,
                                  zoom_in_params=get_zoom_in_params(args.eval_mode, name))
        feats_callback = None
        if args.save_feats and rank == 0 and world == 1:
            from isegprobe_amd.core.inference.utils import get_save_feats_callback
            feats_callback = get_save_feats_callback(logs, name, feats_folder, exec_for_n_imgs=args.save_feats)
        all_ious, elapsed = evaluate_dataset(dataset, predictor, pred_thr=args.thresh, max_iou_thr=max_iou_thr,
                                             feats_callback=feats_callback,
                                             min_clicks=args.min_n_clicks, max_clicks=args.n_clicks,
                                             device_clicker=False if args.host_clicker else None, shard=shard)
        if rank:  # the gathered arrays are on every rank; the table and the log files are rank 0's
            out.append((name, all_ious, None))
            continue
        # the reference's table / log files (inference/utils.py:174-246,365-543); NoC thresholds up to target_iou
        res = save_results(model.upsampler.__class__.__name__, name, logs, (all_ious, elapsed), eval_mode=args.eval_mode,
                           n_clicks=args.n_clicks, target_iou=max_iou_thr if print_ious else args.target_iou, print_ious=print_ious,
                           save_ious=True, print_header=i == 0)
        out.append((name, all_ious, res))
        save_iou_analysis_data(name, logs, (all_ious, elapsed), eval_mode=args.eval_mode, n_clicks=args.n_clicks)
        print(f"{name}: SPC {elapsed / max(sum(len(x) for x in all_ious), 1):.4f} s; logs, IoU pickles: {logs}")
    if world > 1:
        from torch import distributed as dist
        dist.barrier()  # nobody is still reading the synthetic tree when rank 0 removes it
        dist.destroy_process_group()
    if tmp:
        tmp.cleanup()
    return out


if __name__ == "__main__":
    main()
