#!/usr/bin/env python3
"""Benchmark of the per-click dense-feature path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward pass of the whole path over one synthetic batch per GPU:
click maps -> normalise -> DINOv2-S/14 (clicks injected before the blocks) -> FeatUp JBU x16
-> bilinear resize to the image size (fused into the last JBU stage) -> ConvSegHead -> logits   (BASELINE.json configs[1]:
"DINOv2-S/14 + FeatUp JBU, 448x448 batch=32, forward-only").  The metric's "featurizer +
upsampler" stages are inside the timed region together with the seg head and the click-map
generator that north_star places on the same path (more work, never less); their separate
rates are reported under "stages".  Inputs are resident in HBM before the timed region.
Weak scaling: every rank runs its own batch, no data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

S14 = dict(img_size=518, patch_size=14, embed_dim=384, depth=12, num_heads=6)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA


def synthetic_batch(B, S, seed, P=24):
    """image ~ U[0,1], prev mask 0, clicks n_pos~U{1..P}, n_neg~U{0..P}."""
    g = torch.Generator().manual_seed(seed)
    image = torch.rand(B, 4, S, S, generator=g)
    image[:, 3] = 0
    rng = np.random.default_rng(seed)
    pts = -np.ones((B, 2 * P, 3), dtype=np.float32)
    for b in range(B):
        npos, nneg = rng.integers(1, P + 1), rng.integers(0, P + 1)
        k = 0
        for pol, n in ((0, npos), (1, nneg)):
            for i in range(n):
                pts[b, pol * P + i] = (rng.integers(0, S), rng.integers(0, S), k)
                k += 1
    return image, torch.from_numpy(pts)


def build(upsampler, size):
    from helpers import build_model, seeded_
    params = {"backbone_type": "dinov2"} if upsampler == "jbu_featup" else None
    model = build_model(upsampler, vit=S14, img=(size, size), upsampler_params=params)
    seeded_(model, 2025)  # random-init weights of the named architecture (no checkpoints offline)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    return model


class ConvTimer:
    """HIP events around every 3x3-conv launch of the timed steps (same stream as the launch).  The
    seg head issues them through three wrappers (plain, folded-affine first layer, classifier-fused
    last layer); all three run conv3x3_patch4_kernel_192<...> with the same algorithmic FLOPs."""

    NAMES = ("conv3x3", "conv3x3_folded_affine", "conv3x3_relu_classifier")

    def __init__(self, ops):
        self.ops, self.orig, self.pairs, self.flops = ops, {n: getattr(ops, n) for n in self.NAMES}, [], 0.0

    def __enter__(self):
        def make(fn):
            def timed(x, Wt, *a, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = fn(x, Wt, *a, **k)
                e.record()
                self.pairs.append((s, e))
                B, H, W, C = x.shape
                self.flops = 2.0 * B * H * W * C * 9 * Wt.shape[0]  # algorithmic FLOPs of one launch
                return y
            return timed
        for n, fn in self.orig.items():
            setattr(self.ops, n, make(fn))
        return self

    def __exit__(self, *exc):
        for n, fn in self.orig.items():
            setattr(self.ops, n, fn)

    def mean_ms(self):
        return float(np.mean([s.elapsed_time(e) for s, e in self.pairs])) if self.pairs else None


def stage_times(model, image, points, iters=5):
    """Separate HIP-event timings of the stages (outside the headline timed region)."""
    from isegprobe_amd import hip_ops as ops  # noqa: F401

    def ev():
        return torch.cuda.Event(enable_timing=True)

    out = {}
    with torch.no_grad():
        img, prev = model.prepare_input(image)
        maps = model.dist_maps(img, points)
        feats = model.backbone.forward_fused_clicks(img, prev, maps, model.embed_coords)
        hr = model.upsampler(source=feats, guidance=img)

        def timed(fn):
            fn()
            torch.cuda.synchronize()
            s, e = ev(), ev()
            s.record()
            for _ in range(iters):
                fn()
            e.record()
            torch.cuda.synchronize()
            return s.elapsed_time(e) / iters

        out["click_maps+normalize_ms"] = timed(lambda: (model.prepare_input(image), model.dist_maps(img, points)))
        out["featurizer_ms"] = timed(lambda: model.backbone.forward_fused_clicks(img, prev, maps, model.embed_coords))
        stack = getattr(model.upsampler, "upsampler", None)
        if getattr(model, "fold_upsampler_affine", False) and hasattr(stack, "forward_stages"):
            # the route the timed step takes: JBU stages (last one fused with the resize to the image size), the
            # stack's final fix-up folded into the head's first conv
            hr_f = stack.forward_stages(feats, img, out_size=img.shape[2:])
            Wf, bf, alpha = stack.fixup_affine()
            out["upsampler(+resize)_ms"] = timed(lambda: stack.forward_stages(feats, img, out_size=img.shape[2:]))
            out["head_ms"] = timed(lambda: model.head.forward_folded_affine(hr_f, Wf, bf, alpha))
            out["featurizer+upsampler_images_per_sec"] = image.shape[0] / (
                (out["click_maps+normalize_ms"] + out["featurizer_ms"] + out["upsampler(+resize)_ms"]) * 1e-3)
            out["note"] = "same kernels as the timed step; plugin-by-plugin route (unfolded fix-up GEMM, separate resize): " \
                          f"upsampler {timed(lambda: model.upsampler(source=feats, guidance=img)):.2f} ms, " \
                          f"resize+head {timed(lambda: model._resize_and_head(img, hr)):.2f} ms"
        else:
            out["upsampler_ms"] = timed(lambda: model.upsampler(source=feats, guidance=img))
            out["resize+head_ms"] = timed(lambda: model._resize_and_head(img, hr))
    return out


def cpu_baseline(model_sd, size, upsampler, seed):
    """The CPU oracle (kind "port": torch-CPU restatement of the reference path, pinned by the
    golden fixtures) on ONE image of the same workload, all host threads."""
    from oracle import model as omodel
    n = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(n)
    image, points = synthetic_batch(1, size, seed)
    cfg = dict(patch=14, depth=12, heads=6, upsampler=upsampler, injection="before_backbone",
               with_prev_mask=True, use_disks=True, norm_radius=5)
    t0 = time.perf_counter()
    omodel.forward(image, points, model_sd, cfg)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 image ({size}x{size}, batch 1) through the same path, single run of {dt:.1f} s, fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--upsampler", default="jbu_featup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", init_method="env://")  # RCCL

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from isegprobe_amd import hip_ops as ops
    import logging
    logging.getLogger("root").setLevel(logging.WARNING)

    model = build(args.upsampler, args.size)
    sd = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.cuda()
    image, points = synthetic_batch(args.batch, args.size, seed=1000 + rank)
    image, points = image.cuda(), points.cuda()

    with torch.no_grad():
        for _ in range(args.warmup):
            model(image, points)
        barrier()
        with ConvTimer(ops) as ct:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model(image, points)["instances"]
            barrier()
            dt = time.perf_counter() - t0
    assert out.shape == (args.batch, 1, args.size, args.size) and torch.isfinite(out).all()
    if dist is not None:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    if rank == 0:
        conv_ms = ct.mean_ms()
        traffic = None  # PMC passes cannot run inside bench.py: use the committed measurement of this launch shape
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_conv_pmc.json")))
            if args.batch == 32 and args.size == 448:
                traffic = pmc["derived"]["traffic_bytes_per_launch"]
        except Exception:
            pass
        achieved = ct.flops / (conv_ms * 1e-3) / 1e12
        line = {
            "metric": "images/sec thru featurizer+upsampler @448^2 (whole per-click path: click maps, "
                      "featurizer, upsampler, seg head)",
            "value": world * args.batch * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"DINOv2-S/14 + {args.upsampler} + ConvSegHead(384,2,1), "
                                   f"{args.size}x{args.size}, batch {args.batch}/GPU, forward-only, "
                                   "seeded random-init weights", "per_gpu_batch": args.batch,
                       "global_batch": world * args.batch, "image_size": args.size, "parallelism": f"replicas x{world}"},
            "roofline": {"kernel": "conv3x3_patch4_kernel_192 (seg-head 3x3 conv, implicit GEMM, LDS-resident input patch, one wave per SIMD)",
                         "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "fabric bytes per launch from rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE, profiles/r01_conv_pmc.json",
                         "launch_ms": conv_ms, "flops_per_launch": ct.flops},
        }
        try:  # on-box probe of the dense-bf16 ceiling on random operands (profiles/r01_peaks.json; tools/peaks.py)
            pk = json.load(open(os.path.join(ROOT, "profiles", "r01_peaks.json")))["mfma_bf16_16x16x32_register_loop_tflops"]
            line["roofline"]["peak_measured_random_operands"] = pk["random"]
            line["roofline"]["frac_of_measured"] = achieved / pk["random"]
        except Exception:
            pass
        if not args.no_stages:
            try:
                st = stage_times(model, image, points)
                if "upsampler_ms" in st:
                    st["featurizer+upsampler_images_per_sec"] = args.batch / ((st["featurizer_ms"] + st["upsampler_ms"]) * 1e-3)
                line["stages"] = st
            except Exception as exc:  # the headline number must survive a failure of the extras
                line["stages"] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, args.size, args.upsampler, seed=1000)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
