#!/usr/bin/env python3
"""Benchmark of the per-click dense-feature path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5              # BASELINE.json configs[1] (the headline)
    python bench.py --gpus N ...                                # spawns its own N workers (one per GPU, RCCL)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # or under torchrun
    python bench.py --mode train --gpus N                       # configs[2]/[4]: train step with the RCCL gradient all-reduce

forward mode (default).  One "step" = one forward pass of the whole path over one synthetic batch per GPU:
click maps -> normalise -> DINOv2-S/14 (clicks injected before the blocks) -> FeatUp JBU x16 -> bilinear resize to the
image size (fused into the last JBU stage) -> ConvSegHead -> logits (BASELINE.json configs[1]: "DINOv2-S/14 + FeatUp
JBU, 448x448 batch=32, forward-only").  The metric's "featurizer + upsampler" stages are inside the timed region together
with the seg head and the click-map generator that north_star places on the same path (more work, never less).  Inputs are
resident in HBM before the timed region.  Weak scaling: every rank runs its own batch, no data-path collective
(SURVEY.md 8(e)).

train mode.  One "step" = DataParallelTrainer.step on a per-GPU minibatch: forward, NFL loss, HIP backward through the
frozen trunk and upsampler, ONE flat-bucket all-reduce of the trainable gradients (backend "nccl" = RCCL over xGMI), Adam
(reference core/training/trainer.py:193-314, core/utils/distributed.py:66-78).

Rooflines in the JSON line (forward mode), all from HIP events on the launch stream INSIDE the timed steps:
  roofline            the dominant kernel: the seg head's 3x3 conv (MFMA)
  roofline_vit        the DINOv2 blocks (patch embed + 12 blocks + final norm), SURVEY.md 8(d) ViT(D,L,N) FLOPs (MFMA)
  roofline_attention  the fused attention launches alone, L*N*4ND FLOPs over the step (MFMA)
  roofline_upsampler  the upsampler stage (FeatUp JBU: four stages + the fused resize), its algorithmic bytes (HBM)
and, from the same run (never `value`): `loftup448` (the LoftUp upsampler at 448^2 against the MFMA and the HBM roofline) and
`size896` (the same path on an 896 x 896 batch): the other numbers BASELINE.json's north_star names.
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

VITS = {"dinov2_vits14": dict(img_size=518, patch_size=14, embed_dim=384, depth=12, num_heads=6),
        "dinov2_vitb14": dict(img_size=518, patch_size=14, embed_dim=768, depth=12, num_heads=12),
        "dinov2_vitl14": dict(img_size=518, patch_size=14, embed_dim=1024, depth=24, num_heads=16)}
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


def synthetic_train_batch(B, S, seed, P=24, device="cuda"):
    """SBD-shaped train batch: image, one elliptical instance mask, its first positive click (the trainer simulates
    the corrective clicks on the device)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[:S, :S]
    images = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(seed))
    gts, pts = [], -np.ones((B, 2 * P, 3), np.float32)
    for b in range(B):
        cy, cx, ry, rx = rng.uniform(0.3, 0.7) * S, rng.uniform(0.3, 0.7) * S, rng.uniform(0.1, 0.3) * S, rng.uniform(0.1, 0.3) * S
        m = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1).astype(np.float32)
        images[b] += torch.from_numpy(m)[None] * 0.5
        gts.append(torch.from_numpy(m)[None])
        pts[b, 0] = (int(cy), int(cx), 0)
    return {"images": images.clamp(0, 1).to(device), "instances": torch.stack(gts).to(device),
            "points": torch.from_numpy(pts).to(device)}


def upsampler_params(upsampler, dim):
    return {"jbu_featup": {"backbone_type": "dinov2", "feat_dim": dim},
            "loftup": {"upsampler_path": None, "n_dim": dim},
            "lift": {"lift_path": None, "n_dim": dim, "patch": 14}}.get(upsampler)


def build(upsampler, size, arch="dinov2_vits14"):
    from helpers import build_model, seeded_
    vit = VITS[arch]
    model = build_model(upsampler, vit=vit, img=(size, size), upsampler_params=upsampler_params(upsampler, vit["embed_dim"]))
    seeded_(model, 2025)  # random-init weights of the named architecture (no checkpoints offline)
    with torch.no_grad():
        model.backbone.model.pos_embed.mul_(0.3)
    return model


# ---------------------------------------------------------------------------------------------- algorithmic work
def vit_flops(D, L, hw, patch=14, in_ch=6):
    """ViT(D,L,N) = L*N*(24 D^2 + 4 N D) + the image and click patch embeds (2 * 2*588*D*h*w)."""
    N = hw + 1
    return L * N * (24.0 * D * D + 4.0 * N * D) + 2.0 * in_ch * patch * patch * D * hw


def attention_flops(D, L, hw):
    N = hw + 1
    return L * N * 4.0 * N * D


def upsampler_bytes(upsampler, C, h, w, H, W, e=2):
    """Algorithmic HBM bytes per image of the upsampler stage, SURVEY.md 8(d) (e = bytes per element on this path)."""
    if upsampler == "jbu_featup":  # sum over stages: source read + x2 map written + 49-tap kernels; last stage writes H x W
        total, hs, ws = 0.0, h, w
        for s in range(4):
            oh, ow = (2 * hs, 2 * ws) if s < 3 else (H, W)
            total += C * (hs * ws + oh * ow) * e + 49 * (2 * hs) * (2 * ws) * e
            hs, ws = 2 * hs, 2 * ws
        return total + 3 * H * W * 4  # + the guidance image read once
    if upsampler in ("bilinear", "nearest", "bicubic"):
        return C * (h * w + H * W) * e
    if upsampler == "loftup":
        return H * W * (3 * 4 + C * e) + h * w * (C + 20) * e
    if upsampler == "lift":
        return C * (h * w + 4 * h * w) * e + 3 * H * W * 4 + C * H * W * e  # LiFT x2 + the model's resize to H x W
    return None


# ---------------------------------------------------------------------------------------------- clock / power samples
class Telemetry:
    """Shader clock and socket power sampled on a host thread while steps of the workload run (bench.py uses it around the timed
    region, not inside it), so that a few-percent swing of the headline between boxes or rounds is attributable (the head convolutions run at the socket's power limit: the clock
    the chip holds there differs from device to device).  Source: the amdgpu hwmon files of the device torch runs on
    (freq1_input = gfx clock in Hz, power1_average / power1_input in microwatts), else `rocm-smi --json` snapshots.
    Reading a sysfs file costs microseconds and touches neither the GPU queue nor the timed thread."""

    def __init__(self, period=0.1):
        import threading
        self.period, self.samples, self._stop, self._thr = period, [], threading.Event(), None
        self.hwmon, self.source = self._find_hwmon(), None
        self.source = "hwmon:" + self.hwmon if self.hwmon else "rocm-smi"

    @staticmethod
    def _find_hwmon():
        import glob
        cands = []
        try:
            pr = torch.cuda.get_device_properties(torch.cuda.current_device())
            bdf = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            cands += glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:
            pass
        if not cands:
            cands = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
        for c in cands:
            if os.path.exists(os.path.join(c, "freq1_input")):
                return c
        return None

    def _read(self):
        if self.hwmon:
            out = {}
            try:
                out["sclk_mhz"] = int(open(os.path.join(self.hwmon, "freq1_input")).read()) / 1e6
            except Exception:
                pass
            for f in ("power1_average", "power1_input"):
                try:
                    out["power_w"] = int(open(os.path.join(self.hwmon, f)).read()) / 1e6
                    break
                except Exception:
                    continue
            return out or None
        try:
            import subprocess
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
            card = next(iter(json.loads(r.stdout).values()))
            out = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "mhz" in str(v).lower():
                    out["sclk_mhz"] = float(str(v).lower().replace("(", "").replace(")", "").replace("mhz", ""))
                elif "power" in kl and "(w)" in kl:
                    try:
                        out["power_w"] = float(v)
                    except Exception:
                        pass
            return out or None
        except Exception:
            return None

    def __enter__(self):
        import threading

        def loop():
            while not self._stop.is_set():
                v = self._read()
                if v:
                    self.samples.append(v)
                self._stop.wait(self.period if self.hwmon else 2.0)
        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thr.join(timeout=30)

    def summary(self):
        out = {"source": self.source, "samples": len(self.samples)}
        for key in ("sclk_mhz", "power_w"):
            v = [s[key] for s in self.samples if key in s]
            if v:
                out[key] = {"min": float(np.min(v)), "mean": float(np.mean(v)), "max": float(np.max(v))}
        return out


# ---------------------------------------------------------------------------------------------- in-region HIP-event timers
class OpTimer:
    """HIP events (on the launch stream) around every call of the named callables during the timed steps."""

    def __init__(self, owner, names):
        self.owner, self.names = owner, [n for n in names if hasattr(owner, n)]
        self.orig, self.pairs, self.last_args = {n: getattr(owner, n) for n in self.names}, [], None

    def __enter__(self):
        def make(fn):
            def timed(*a, **k):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                y = fn(*a, **k)
                e.record()
                self.pairs.append((s, e))
                self.last_args = a
                return y
            return timed
        for n, fn in self.orig.items():
            setattr(self.owner, n, make(fn))
        return self

    def __exit__(self, *exc):
        for n, fn in self.orig.items():
            setattr(self.owner, n, fn)

    def times_ms(self):
        return [s.elapsed_time(e) for s, e in self.pairs]

    def mean_ms(self):
        t = self.times_ms()
        return float(np.mean(t)) if t else None


def stage_times(model, image, points, iters=5):
    """Separate, sequential HIP-event timings of the stages (outside the headline timed region)."""
    from isegprobe_amd import hip_ops as ops

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
        vm = model.backbone.model
        heads, D = vm.num_heads, vm.embed_dim
        Bn, L = image.shape[0], feats.shape[2] * feats.shape[3] + 1
        qkv = torch.randn(Bn * L, 3 * D, device=image.device).to(torch.bfloat16)
        out["attention_launch_ms"] = timed(lambda: ops.attention_packed_qkv(qkv, Bn, L, heads, (D // heads) ** -0.5))
        del qkv
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
        else:
            out["upsampler_ms"] = timed(lambda: model.upsampler(source=feats, guidance=img))
            out["resize+head_ms"] = timed(lambda: model._resize_and_head(img, hr))
            out["featurizer+upsampler_images_per_sec"] = image.shape[0] / (
                (out["click_maps+normalize_ms"] + out["featurizer_ms"] + out["upsampler_ms"]) * 1e-3)
    return out


# ---------------------------------------------------------------------------------------------- north_star's other named numbers
def loftup_flops(C, HW, hw):
    """LoftUp(C, HW, hw), c = C + 20: HW*[18*203*c + 18c^2 + 2*(4c^2 + 4*hw*c + 4cC) + 2cC] FLOP."""
    c = C + 20
    return HW * (18.0 * 203 * c + 18.0 * c * c + 2 * (4.0 * c * c + 4.0 * hw * c + 4.0 * c * C) + 2.0 * c * C)


def loftup448_block(B=8, C=384, S=448, warm=2, iters=5):
    """north_star: "LoftUp upsampler at 448^2" against BOTH rooflines (contraction-bound as an algorithm,
    HBM-bound only in the reference's materialised-attention form).  The upsampler plugin alone on a batch of B images:
    source [B,C,32,32] tokens + guidance image -> [B,C,448,448]."""
    from helpers import seeded_
    from isegprobe_amd.core.model.upsamplers import LoftUpUpsampler
    up = seeded_(LoftUpUpsampler(None, n_dim=C), 3).cuda().eval()
    h = S // 14
    src = torch.randn(B, C, h, h, device="cuda")
    gd = torch.randn(B, 3, S, S, device="cuda")
    with torch.no_grad():
        for _ in range(warm):
            y = up(src, gd)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            y = up(src, gd)
        e.record()
        torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    assert y.shape == (B, C, S, S) and torch.isfinite(y.float()).all()
    fl, by = B * loftup_flops(C, S * S, h * h), B * upsampler_bytes("loftup", C, h, h, S, S)
    tf, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
    del up, y
    torch.cuda.empty_cache()
    return {"workload": f"LoftUp(n_dim={C}) upsampler plugin alone, {S}x{S}, batch {B}, {h}x{h} LR tokens, inference stream (IEEE half)",
            "ms_per_batch": ms, "ms_per_image": ms / B, "images_per_sec": B / (ms * 1e-3),
            "mfma": {"achieved": tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_PEAK_TFLOPS, "flops_per_batch": fl},
            "min_bytes_per_batch": by,
            "note": "contraction-bound as an algorithm (SURVEY.md 8(d)): its minimal traffic, HW*(3*4 + 2C) + hw*(C+20)*2 bytes per image, "
                    f"is {gbs:.0f} GB/s at this rate -- HBM is not a bound here, so no HBM fraction is quoted (north_star's '60 % of HBM peak' "
                    "describes the reference's materialised attention weights, 3.3 GB per image and layer)"}


def size896_block(arch, upsampler, B=8, S=896, warm=2, iters=5):
    """north_star: "images/sec on synthetic ... 896^2 batches": the same per-click path at 896 x 896 (64 x 64 tokens, N = 4097)."""
    from isegprobe_amd import hip_ops as ops
    vit = VITS[arch]
    model = build(upsampler, S, arch).cuda()
    image, points = synthetic_batch(B, S, seed=896)
    image, points = image.cuda(), points.cuda()
    with torch.no_grad():
        for _ in range(warm):
            model(image, points)
        torch.cuda.synchronize()
        with OpTimer(ops, ("conv3x3", "conv3x3_folded_affine", "conv3x3_relu_classifier")) as t_conv, \
                OpTimer(model.backbone, ("forward_fused_clicks",)) as t_vit:
            t0 = time.perf_counter()
            for _ in range(iters):
                out = model(image, points)["instances"]
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    assert out.shape == (B, 1, S, S) and torch.isfinite(out).all()
    D, L, hw = vit["embed_dim"], vit["depth"], (S // 14) ** 2
    conv_ms, vit_ms = t_conv.mean_ms(), t_vit.mean_ms()
    xin, Wt = t_conv.last_args[0], t_conv.last_args[1]
    conv_fl = 2.0 * xin.shape[0] * xin.shape[1] * xin.shape[2] * xin.shape[3] * 9 * Wt.shape[0]
    vfl = B * vit_flops(D, L, hw)
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    del model, out
    torch.cuda.empty_cache()
    return {"workload": f"{arch} + {upsampler} + ConvSegHead({D},2,1), {S}x{S}, batch {B}, forward-only", "images_per_sec": B * iters / dt,
            "ms_per_step": dt / iters * 1e3, "steps": iters, "warmup": warm,
            "head_conv": {"launch_ms": conv_ms, "achieved": conv_fl / (conv_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                          "frac": conv_fl / (conv_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS},
            "vit": {"ms_in_step": vit_ms, "achieved": vfl / (vit_ms * 1e-3) / 1e12, "unit": "TFLOP/s",
                    "frac_in_step": vfl / (vit_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                    "attention_share_of_flops": attention_flops(D, L, hw) / vit_flops(D, L, hw)},
            "peak_mem_GiB": mem}


def cfg3_block(S=896, warm=2, iters=5):
    """BASELINE configs[3]: "ViT-L/14 + LiFT, 896x896 high-res click loop": one click of the loop = the image and its mirrored
    copy (batch 2, predictor flip TTA) through DINOv2-L/14 + LiFT(1024) + ConvSegHead(1024,2,1), forward only."""
    arch = "dinov2_vitl14"
    vit = VITS[arch]
    torch.cuda.reset_peak_memory_stats()
    model = build("lift", S, arch).cuda()
    image, points = synthetic_batch(2, S, seed=3896)
    image, points = image.cuda(), points.cuda()
    with torch.no_grad():
        for _ in range(warm):
            model(image, points)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(image, points)["instances"]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert out.shape == (2, 1, S, S) and torch.isfinite(out).all()
    D, L, hw = vit["embed_dim"], vit["depth"], (S // 14) ** 2
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    # the same click with LiFT's 128 x 128 map resized to 896 x 896 and convolved there (rounds 1-3; 1.6 GB per image in 16 bits)
    from isegprobe_amd.core.model.heads import conv_heads
    saved = conv_heads.CONV_OF_BILINEAR
    conv_heads.CONV_OF_BILINEAR = False
    try:
        torch.cuda.reset_peak_memory_stats()
        dt_old, out_old = _time_forward(model, image, points, 1, 3)
        mem_old = torch.cuda.max_memory_allocated() / 2 ** 30
    finally:
        conv_heads.CONV_OF_BILINEAR = saved
    diff = (out - out_old).abs().max().item()
    del model, out, out_old
    torch.cuda.empty_cache()
    return {"workload": f"{arch} + lift + ConvSegHead({D},2,1), {S}x{S}, batch 2 (image + mirrored copy = one click), forward-only; "
                        "first head convolution through the resize (low-resolution GEMM + blend)",
            "ms_per_click": dt / iters * 1e3, "clicks_per_sec": iters / dt, "vit_flops_per_click": 2 * vit_flops(D, L, hw),
            "steps": iters, "warmup": warm, "peak_mem_GiB": mem,
            "materialised_route": {"ms_per_click": dt_old * 1e3, "peak_mem_GiB": mem_old, "max_abs_logit_diff_vs_default": diff}}


def _time_forward(model, image, points, warm, iters):
    with torch.no_grad():
        for _ in range(warm):
            model(image, points)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model(image, points)["instances"]
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters, out


def cfg0_block(B=32, S=448, warm=2, iters=5):
    """BASELINE configs[0]'s model on the GPU: DINOv2-S/14 + bilinear upsampler + ConvSegHead(384,2,1), forward only.  The head's
    first convolution runs THROUGH the bilinear plugin's resize (one [B*h*w, C] x [C, 9N] GEMM at low resolution + the blend
    kernel, csrc/conv_bilinear.hip); `materialised_route` is the same step with the [B,S,S,C] map written and convolved
    (rounds 1-3, ISEGPROBE_CONV_OF_BILINEAR=0)."""
    from isegprobe_amd import hip_ops as ops
    from isegprobe_amd.core.model.heads import conv_heads
    arch = "dinov2_vits14"
    D = VITS[arch]["embed_dim"]
    model = build("bilinear", S, arch).cuda()
    image, points = synthetic_batch(B, S, seed=448)
    image, points = image.cuda(), points.cuda()
    with OpTimer(ops, ("conv3x3_of_bilinear_blend",)) as t_blend:
        dt, out = _time_forward(model, image, points, warm, iters)
    assert out.shape == (B, 1, S, S) and torch.isfinite(out).all() and t_blend.pairs, "the through-the-resize route did not run"
    blend_ms = float(np.mean(t_blend.times_ms()[-iters:]))
    saved = conv_heads.CONV_OF_BILINEAR
    conv_heads.CONV_OF_BILINEAR = False
    try:
        dt_old, out_old = _time_forward(model, image, points, warm, iters)
    finally:
        conv_heads.CONV_OF_BILINEAR = saved
    diff = (out - out_old).abs().max().item()
    h = S // 14
    del model, out, out_old
    torch.cuda.empty_cache()
    out_bytes = B * S * S * D * 2.0
    return {"workload": f"{arch} + bilinear + ConvSegHead({D},2,1), {S}x{S}, batch {B}, forward-only (BASELINE configs[0]'s model on the GPU)",
            "images_per_sec": B / dt, "ms_per_step": dt * 1e3, "steps": iters, "warmup": warm,
            "first_conv_through_resize": {"blend_launch_ms": blend_ms, "bound": "hbm", "algorithmic_bytes": out_bytes + B * h * h * 9 * D * 2.0,
                                          "achieved": (out_bytes + B * h * h * 9 * D * 2.0) / (blend_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": (out_bytes + B * h * h * 9 * D * 2.0) / (blend_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                          "write_ceiling_note": "on-box ceilings for a written map of this size (tools/write_bw.py): broadcast multiply 3.8 TB/s, "
                                                                "the plain resize kernel writing the same map 2.9 TB/s, a constant fill 6.9 TB/s",
                                          "note": "blend kernel (both separable phases on v_mfma_f32_16x16x16_f16 fed by transposed LDS reads): writes the "
                                                  "[B,S,S,N] half map once, reads the [B*h*w, 9N] tap planes"},
            "materialised_route": {"images_per_sec": B / dt_old, "ms_per_step": dt_old * 1e3, "max_abs_logit_diff_vs_default": diff}}


def fp32_mode_block(arch, upsampler, B=32, S=448, iters=2):
    """The NoC-identical mode (`evaluate.py --fp32`, core/model/precise.py: every contraction as three bf16 products with fp32
    accumulation, everything between them in fp32) on the HEADLINE workload: what the configuration whose NoC equals the
    reference's per object costs per step, next to the 16-bit path the headline is measured on.  `noc_16bit_vs_reference`
    states what the 16-bit path gives up on the three dataset fixtures (tests/test_noc_dataset_gpu.py holds both)."""
    model = build(upsampler, S, arch).cuda()
    image, points = synthetic_batch(B, S, seed=1000)
    image, points = image.cuda(), points.cuda()
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        ref16 = model(image, points)["instances"]
        model.forward_fp32(image, points)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            out = model.forward_fp32(image, points)["instances"]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
        dt16, _ = _time_forward(model, image, points, 1, 3)
    diff = (out - ref16).abs()
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    del model, out, ref16
    torch.cuda.empty_cache()
    return {"workload": f"{arch} + {upsampler} + ConvSegHead, {S}x{S}, batch {B}, forward_fp32 (fp32-accurate: three bf16 products per contraction)",
            "ms_per_step": dt * 1e3, "images_per_sec": B / dt, "x_16bit_path": dt / dt16, "ms_per_step_16bit_same_model": dt16 * 1e3,
            "logits_16bit_minus_fp32_mode": {"max": diff.max().item(), "rms": diff.pow(2).mean().sqrt().item()},
            "peak_mem_GiB": mem, "steps": iters,
            "noc_16bit_vs_reference": {
                "source": "tests/test_noc_dataset_gpu.py over tests/golden/noc_dataset*.npz (reference NoBRS evaluation, 20 clicks, flip + zoom-in)",
                "grabcut_layout_bilinear_50_objects": {"NoC@80/85/90": {"reference = fp32 mode": [5.82, 8.22, 12.26], "16-bit path": [5.74, 8.20, 11.96]}, "objects_differing": 4},
                "grabcut_layout_lift": {"NoC@85": {"reference = fp32 mode": 4.10, "16-bit path": 4.00}, "objects_differing": 1},
                "grabcut_layout_loftup": {"NoC@80/85/90": {"reference = fp32 mode = 16-bit path": [2.68, 3.88, 5.84]}, "objects_differing": 0},
                "sbd_layout_46_objects": {"NoC@80/85/90": {"reference = fp32 mode": [12.07, 17.07, 19.72], "16-bit path": [12.07, 17.09, 19.63]}, "objects_differing": 2},
                "measured": "round 4, with the head's first convolution through the resize on the 16-bit path (python -m pytest tests/test_noc_dataset_gpu.py -s)"}}


def train_block(arch, upsampler, B, S, sim_clicks=2, warm=2, iters=5):
    """BASELINE configs[2] / [4] on this GPU's share: the SBD-shaped train step (clicks before the backbone, `sim_clicks` no-grad
    simulated-click forwards, train-mode forward, HIP backward, flat-bucket all-reduce -- a no-op in one process -- and Adam).
    `python bench.py --mode train` is the data-parallel measurement; this block only puts the single-GPU number into the line."""
    from isegprobe_amd.core.training.trainer import DataParallelTrainer
    torch.manual_seed(0)
    torch.cuda.reset_peak_memory_stats()
    model = build(upsampler, S, arch).cuda()
    trainer = DataParallelTrainer(model, lr=5e-5)
    batch = synthetic_train_batch(B, S, seed=2000)
    for _ in range(warm):
        trainer.step(batch, num_iters=sim_clicks)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        loss = trainer.step(batch, num_iters=sim_clicks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert torch.isfinite(loss)
    mem, nbytes = torch.cuda.max_memory_allocated() / 2 ** 30, trainer.bucket.nbytes()
    del trainer, model, batch
    torch.cuda.empty_cache()
    return {"workload": f"{arch} + {upsampler} + ConvSegHead, {S}x{S} crops, batch {B}, clicks before the backbone, {sim_clicks} simulated "
                        "corrective clicks per step, one GPU (no collective)", "ms_per_step": dt / iters * 1e3,
            "images_per_sec": B * iters / dt, "gradient_bucket_bytes": nbytes, "steps": iters, "warmup": warm, "peak_mem_GiB": mem}


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle)
def cpu_baseline(model_sd, size, upsampler, vit, seed, quick=False):
    """The CPU oracle (kind "port": torch-CPU restatement of the reference path, pinned by the golden fixtures) on
    a bounded sample of the same workload, on this box's host cores: batch 1 (median of 3 runs, per-stage split) and
    batch 8 (median of 3 runs; one with --cpu-baseline-quick), SURVEY.md 8(d).  A reported baseline, not the target."""
    import torch.nn.functional as F
    from oracle import model as omodel
    from oracle import upsamplers as ups
    from oracle import vit as ovit
    from oracle.click_maps import click_maps
    n = min(16, os.cpu_count() or 1)  # the GPU box's CPU share for one GPU is 16 cores
    torch.set_num_threads(n)

    def one(B):
        image, points = synthetic_batch(B, size, seed)
        t = [time.perf_counter()]
        with torch.no_grad():
            img = omodel.normalize(image[:, :3].float())
            maps = torch.from_numpy(click_maps(points.numpy(), size, size, 5, 1.0, True))
            coord = torch.cat((image[:, 3:], maps), 1)
            t.append(time.perf_counter())
            clicks = ovit.patch_tokens(coord, model_sd["embed_coords.proj.weight"], model_sd["embed_coords.proj.bias"], 14)
            feats = ovit.dinov2_features(img, model_sd, patch=14, depth=vit["depth"], heads=vit["num_heads"],
                                         click_tokens=clicks, injection="before_backbone", prefix="backbone.model.")
            t.append(time.perf_counter())
            if upsampler == "jbu_featup":
                hr = ups.jbu_stack(feats, img, model_sd, "upsampler.upsampler.")
            elif upsampler == "loftup":
                hr = ups.loftup(feats, img, model_sd, "upsampler.upsampler.")
            elif upsampler == "lift":
                hr = ups.lift(feats, img, model_sd, "upsampler.lift.")
            else:
                hr = getattr(ups, upsampler)(feats, img)
            if upsampler != "identity" and hr.shape[2:] != img.shape[2:]:
                hr = F.interpolate(hr, size=img.shape[2:], mode="bilinear", align_corners=True)
            t.append(time.perf_counter())
            logits = omodel.conv_head(hr, model_sd)
            F.interpolate(logits, size=img.shape[2:], mode="bilinear", align_corners=True)
            t.append(time.perf_counter())
        return np.diff(t)  # click maps + normalise, featurizer, upsampler (+resize), head

    r1 = np.array([one(1) for _ in range(3)])
    tot1 = r1.sum(1)
    med = int(np.argsort(tot1)[1])
    r8 = np.array([one(8) for _ in range(1 if quick else 3)])
    tot8 = np.sort(r8.sum(1))[len(r8) // 2]
    names = ("click_maps+normalize", "featurizer", "upsampler(+resize)", "head")
    return {"value": 1.0 / tot1[med], "unit": "images/sec", "cores": torch.get_num_threads(), "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"batch 1: median of 3 runs of one {size}x{size} image through the same path ({tot1[med]:.1f} s each, fp32); "
                      f"batch 8: {'one run' if quick else 'median of 3 runs'} ({tot8:.1f} s)",
            "batch1_runs_s": [round(float(v), 2) for v in tot1],
            "batch1_stage_s": {k: round(float(v), 3) for k, v in zip(names, r1[med])},
            "batch8_runs_s": [round(float(v), 2) for v in r8.sum(1)],
            "batch8_images_per_sec": 8.0 / float(tot8),
            "batch8_stage_s": {k: round(float(v), 3) for k, v in zip(names, r8[int(np.argsort(r8.sum(1))[len(r8) // 2])])}}


# ---------------------------------------------------------------------------------------------- workers
def _dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev = "cpu" if args.dry_run else "cuda"
    # Rehearsal of the N > 1 paths on a ONE-GPU box: ISEGPROBE_SHARE_GPU=1 puts every rank on device 0 and
    # ISEGPROBE_DIST_BACKEND=gloo replaces RCCL (which wants one device per rank) -- same code, same barriers, same JSON.
    share = os.environ.get("ISEGPROBE_SHARE_GPU", "0") == "1"
    backend = os.environ.get("ISEGPROBE_DIST_BACKEND") or ("gloo" if args.dry_run else "nccl")  # "nccl" = RCCL
    if not args.dry_run:
        torch.cuda.set_device(0 if share else local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, init_method="env://")

    def barrier():
        if dist is not None:
            dist.barrier()
        if not args.dry_run:
            torch.cuda.synchronize()

    def max_over_ranks(dt):
        if dist is None:
            return dt
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.item()
    return world, rank, dist, barrier, max_over_ranks


def _pmc_traffic(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def _head_w_bits(model, fused_jbu):
    """Significant bits of the seg head's half-format weights in the timed steps (ISEGPROBE_HEAD_W_BITS; 8 = bf16-valued
    numbers stored as half, 11 = full half): part of the precision the headline was measured at."""
    from isegprobe_amd.core.model.heads import conv_heads
    return conv_heads.HEAD_W_BITS if (getattr(model, "head_f16", False) and fused_jbu) else None


def run_dry(args):
    """Launch plumbing only (CPU, gloo): the ranks rendezvous, time an empty region the way the real modes do, and
    rank 0 prints one JSON line.  Used by tests/test_distributed_cpu.py to cover `bench.py --gpus N` self-spawn."""
    world, rank, dist, barrier, max_over_ranks = _dist_setup(args)
    barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "mode": args.mode, "max_dt": dt}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_forward(args):
    world, rank, dist, barrier, max_over_ranks = _dist_setup(args)
    from isegprobe_amd import hip_ops as ops
    import logging
    logging.getLogger("root").setLevel(logging.WARNING)

    vit = VITS[args.arch]
    model = build(args.upsampler, args.size, args.arch)
    sd = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.cuda()
    image, points = synthetic_batch(args.batch, args.size, seed=1000 + rank)
    image, points = image.cuda(), points.cuda()
    stack = getattr(model.upsampler, "upsampler", None)
    fused_jbu = getattr(model, "fold_upsampler_affine", False) and hasattr(stack, "forward_stages")

    # Shader clock / socket power are sampled AROUND the timed region -- during the warm-up steps in front of it and during a few
    # untimed steps of the same workload behind it -- never inside it: a sampler thread polling the SMU through hwmon while the
    # timed steps ran cost one run 10 % of its headline (823 against 934 img/s for the region that followed without it).
    with torch.no_grad():
        with Telemetry(period=0.05) as tele_pre:
            for _ in range(args.warmup):
                model(image, points)
            barrier()
        with OpTimer(ops, ("conv3x3", "conv3x3_folded_affine", "conv3x3_relu_classifier")) as t_conv, \
                OpTimer(ops, ("attention_packed_qkv",)) as t_att, \
                OpTimer(model.backbone, ("forward_fused_clicks",)) as t_vit, \
                OpTimer(stack if fused_jbu else model.upsampler, ("forward_stages",) if fused_jbu else ("forward",)) as t_up:
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out = model(image, points)["instances"]
            barrier()
            dt = time.perf_counter() - t0
        with Telemetry(period=0.05) as tele_post:
            for _ in range(min(5, args.steps)):
                model(image, points)
            barrier()
    assert out.shape == (args.batch, 1, args.size, args.size) and torch.isfinite(out).all()
    dt = max_over_ranks(dt)

    # The same steps with the head's convolutions on bf16 instead of IEEE-half operands (ISEGPROBE_HEAD_F16=0): reported
    # beside the headline as `alt_head_bf16`, never as `value` -- the half form costs ~3 % (its multipliers draw more power
    # and the socket is at its limit during these kernels) and buys the logit-error margin DESIGN.md section 4 describes.
    dt_bf16 = None
    if getattr(model, "head_f16", False) and fused_jbu and world == 1 and not args.no_alt:
        from isegprobe_amd.core.model.heads import conv_heads
        saved = (model.head_f16, conv_heads.HEAD_F16)
        model.head_f16 = conv_heads.HEAD_F16 = False
        try:
            with torch.no_grad():
                for _ in range(max(2, args.warmup // 2)):
                    model(image, points)
                barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    model(image, points)
                barrier()
                dt_bf16 = time.perf_counter() - t0
        finally:
            model.head_f16, conv_heads.HEAD_F16 = saved

    # The same steps with FeatUp-JBU's guidance-only records on the MAIN stream (ISEGPROBE_JBU_SIDE_STREAM=0): what the overlap on
    # the second stream buys, and the trunk's in-step time when nothing shares the chip with it.  Never `value`.
    alt_serial = None
    if fused_jbu and world == 1 and not args.no_alt:
        saved_env = os.environ.get("ISEGPROBE_JBU_SIDE_STREAM")
        os.environ["ISEGPROBE_JBU_SIDE_STREAM"] = "0"
        try:
            with torch.no_grad():
                for _ in range(2):
                    model(image, points)
                barrier()
                with OpTimer(model.backbone, ("forward_fused_clicks",)) as t_vit_serial:
                    t0 = time.perf_counter()
                    for _ in range(args.steps):
                        model(image, points)
                    barrier()
                    dts = time.perf_counter() - t0
            alt_serial = (dts, t_vit_serial.mean_ms())
        finally:
            if saved_env is None:
                os.environ.pop("ISEGPROBE_JBU_SIDE_STREAM", None)
            else:
                os.environ["ISEGPROBE_JBU_SIDE_STREAM"] = saved_env

    if rank == 0:
        B, S, D, L = args.batch, args.size, vit["embed_dim"], vit["depth"]
        h = w = S // 14
        conv_ms = t_conv.mean_ms()
        xin, Wt = t_conv.last_args[0], t_conv.last_args[1]
        conv_flops = 2.0 * xin.shape[0] * xin.shape[1] * xin.shape[2] * xin.shape[3] * 9 * Wt.shape[0]
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12
        pmc = _pmc_traffic("r04_conv_pmc.json") or _pmc_traffic("r03_conv_pmc.json") or _pmc_traffic("r02_conv_pmc.json") or _pmc_traffic("r01_conv_pmc.json")
        traffic = None
        if pmc is not None and B == 32 and S == 448 and args.arch == "dinov2_vits14":
            traffic = pmc["derived"]["traffic_bytes_per_launch"]
        line = {
            "metric": "images/sec thru featurizer+upsampler @448^2 (whole per-click path: click maps, "
                      "featurizer, upsampler, seg head)",
            "value": world * B * args.steps / dt,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not (getattr(model, "head_f16", False) and fused_jbu) else
                     "f16 (IEEE-half 16-bit operands of the ViT blocks, the FeatUp-JBU stack and the seg-head convolutions -- head weights hold `head_w_bits` significant bits; bf16 patch matrix), fp32 accumulation",
            "data": "synthetic",
            "head_w_bits": _head_w_bits(model, fused_jbu),
            "config": {"workload": f"{args.arch} + {args.upsampler} + ConvSegHead({D},2,1), {S}x{S}, batch {B}/GPU, "
                                   "forward-only, seeded random-init weights", "per_gpu_batch": B,
                       "global_batch": world * B, "image_size": S, "parallelism": f"replicas x{world}"},
            "roofline": {"kernel": "conv3x3_patch4_kernel_192 (seg-head 3x3 conv, implicit GEMM, LDS-resident input patch, "
                                   "one wave per SIMD; both launches of the step: folded-affine first conv, classifier-fused second)",
                         "bound": "mfma", "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_note": "fabric bytes per launch from rocprofv3 FETCH_SIZE(x2)+WRITE_SIZE, separate --pmc passes "
                                         "(profiles/r04_conv_pmc.json, else r03 / r02 / r01)",
                         "launch_ms": conv_ms, "flops_per_launch": conv_flops},
        }
        if dt_bf16 is not None:
            line["alt_head_bf16"] = {"value": B * args.steps / dt_bf16, "unit": "images/sec", "ms_per_step": dt_bf16 / args.steps * 1e3,
                                     "note": "same steps with ISEGPROBE_HEAD_F16=0 (bf16 head convolutions: bench-workload logit "
                                             "error 8.5e-3 max / 1.7e-3 rms instead of 5.7e-3 / 1.2e-3)"}
        line["telemetry_around_timed_region"] = {"before_warmup_steps": tele_pre.summary(), "after_untimed_steps": tele_post.summary(),
                                                 "note": "same workload, sampled outside the timed steps (hwmon polling inside them perturbs the run)"}
        if traffic is not None:
            # mean over the step's two launches, like `traffic`: both read a [B,S,S,C] 16-bit map once, the first writes one
            # (the second's output is the classifier's partial sums, 4 bytes x slots per pixel)
            alg = B * S * S * (2.0 * xin.shape[3] * 2 + Wt.shape[0] * 2 + 4.0 * ops._lib.lib().isp_conv3x3_partial_slots(Wt.shape[0])) / 2
            line["roofline"]["algorithmic_bytes_per_launch"] = alg
            line["roofline"]["traffic_over_algorithmic"] = traffic / alg
        if alt_serial is not None:
            fl_v = B * vit_flops(D, L, h * w)
            line["alt_records_on_main_stream"] = {
                "value": B * args.steps / alt_serial[0], "unit": "images/sec", "ms_per_step": alt_serial[0] / args.steps * 1e3,
                "vit_ms_in_step": alt_serial[1], "vit_frac_in_step": fl_v / (alt_serial[1] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                "note": "same steps with ISEGPROBE_JBU_SIDE_STREAM=0: the trunk runs with the chip to itself (its in-step time is its "
                        "stand-alone time) and the step is slower -- the second stream hides the record kernels behind the trunk at the "
                        "price of the trunk's own kernels sharing CUs with them"}
        pk = _pmc_traffic("r01_peaks.json")
        if pk is not None:
            rnd = pk["mfma_bf16_16x16x32_register_loop_tflops"]["random"]
            line["roofline"]["peak_measured_random_operands"] = rnd
            line["roofline"]["frac_of_measured"] = achieved / rnd
        # --- the rooflines north_star names.  Inside the step FeatUp-JBU's guidance-only kernels run beside the ViT on a
        # second stream, so the ViT's own events there measure a shared chip; the roofline uses the stage run on its own
        # (stages.*, same kernels, same inputs) and quotes the in-step event time next to it.
        st = None
        if not args.no_stages:
            try:
                st = stage_times(model, image, points)
                line["stages"] = st
            except Exception as exc:  # the headline number must survive a failure of the extras
                line["stages"] = {"error": repr(exc)}
        vit_in, att_in = t_vit.mean_ms(), t_att.times_ms()
        vit_ms = (st or {}).get("featurizer_ms") or vit_in
        if vit_ms:
            fl = B * vit_flops(D, L, h * w)
            line["roofline_vit"] = {"kernel": "DINOv2 forward: patch embed (image + clicks), blocks, final norm "
                                              "(gemm_tile_kernel, attention64_kernel, layernorm)",
                                    "bound": "mfma", "achieved": fl / (vit_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                                    "unit": "TFLOP/s", "frac": fl / (vit_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                    "ms_per_step": vit_ms, "ms_in_step_beside_jbu_records": vit_in, "flops_per_step": fl,
                                    "frac_in_step": (fl / (vit_in * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS) if vit_in else None,
                                    "note": "frac = the stage run on its own; frac_in_step = HIP events inside the timed steps, where "
                                            "FeatUp-JBU's guidance-only kernels share the chip on a second stream",
                                    "traffic": None}
        trunk_pmc = _pmc_traffic("r04_bench_pmc.json")
        if trunk_pmc is not None and "roofline_vit" in line and B == 32 and S == 448 and args.arch == "dinov2_vits14":
            # fabric bytes per step of the trunk's kernels (4 profiled forwards: 1 warm-up + 3 steps)
            per = {k: v["traffic_bytes_per_launch"] * v["FETCH_SIZE"]["launches"] / 4.0 for k, v in trunk_pmc["kernels"].items()
                   if k.startswith(("gemm_tile_kernel", "attention64", "layernorm_kernel", "patchify_rows")) and "traffic_bytes_per_launch" in v}
            line["roofline_vit"]["traffic"] = sum(per.values())
            line["roofline_vit"]["traffic_note"] = ("fabric bytes per step summed over the trunk's kernels (GEMMs, attention, LayerNorm, patch matrix), "
                                                    "rocprofv3 --pmc passes of this program (profiles/r04_bench_pmc.json)")
            att = [v for k, v in trunk_pmc["kernels"].items() if k.startswith("attention64") and "traffic_bytes_per_launch" in v]
            if att:
                line["_attention_traffic_per_launch"] = att[0]["traffic_bytes_per_launch"]
        att_ms = (st or {}).get("attention_launch_ms") or (float(np.mean(att_in)) if att_in else None)
        if att_ms:
            fl = B * attention_flops(D, L, h * w) / L
            line["roofline_attention"] = {"kernel": "attention64_kernel<true> (fused softmax(QK^T)V on base-2 logits, LDS-staged K/V tiles), one launch per block",
                                          "bound": "mfma", "achieved": fl / (att_ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS,
                                          "unit": "TFLOP/s", "frac": fl / (att_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS,
                                          "launch_ms": att_ms, "launch_ms_in_step_beside_jbu_records": float(np.mean(att_in)) if att_in else None,
                                          "launches_per_step": L, "flops_per_launch": fl, "traffic": line.pop("_attention_traffic_per_launch", None)}
        line.pop("_attention_traffic_per_launch", None)
        up_ms_in = t_up.mean_ms()
        up_ms_seq = None if not st else st.get("upsampler(+resize)_ms", st.get("upsampler_ms"))
        by = upsampler_bytes(args.upsampler, D, h, w, S, S)
        if by is not None and (up_ms_seq or up_ms_in):
            ms = up_ms_seq or up_ms_in
            gbs = B * by / (ms * 1e-3) / 1e9
            up_traffic = None
            allpmc = _pmc_traffic("r04_bench_pmc.json") or _pmc_traffic("r03_bench_pmc.json") or _pmc_traffic("r02_bench_pmc.json")
            if allpmc is not None and args.upsampler == "jbu_featup" and B == 32 and S == 448 and args.arch == "dinov2_vits14":
                # fabric bytes per step of the stage's kernels (4 profiled forwards: 1 warm-up + 3 steps)
                up_traffic = sum(v["traffic_bytes_per_launch"] * v["FETCH_SIZE"]["launches"] / 4.0
                                 for k, v in allpmc["kernels"].items()
                                 if k.startswith(("jbu_", "adaptive_avg_pool", "bf16_to_f16")) and "traffic_bytes_per_launch" in v)
            line["roofline_upsampler"] = {"kernel": f"{args.upsampler} stage incl. the resize to the image size "
                                                    "(jbu_kernels / jbu_apply / jbu_apply_resized, range proj, pooling)",
                                          "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                          "frac": gbs / HBM_PEAK_GBS, "ms_per_step": ms, "bytes_per_step": B * by,
                                          "traffic": up_traffic,
                                          "traffic_note": "fabric bytes per step summed over the stage's kernels, rocprofv3 --pmc "
                                                          "passes of this program (profiles/r04_bench_pmc.json, else r03 / r02)",
                                          "note": "time = the stage run on its own (stages.*): inside the step its guidance-only half "
                                                  f"overlaps the ViT on a second stream (main-stream share {up_ms_in:.2f} ms)"
                                          if up_ms_seq and up_ms_in else "HIP events inside the timed steps"}
        if world == 1 and not args.no_extras:
            # the other numbers north_star names, measured in the same run (never `value`)
            del out
            torch.cuda.empty_cache()
            for key, fn in (("loftup448", lambda: loftup448_block()), ("size896", lambda: size896_block(args.arch, args.upsampler)),
                            ("cfg0_bilinear448", lambda: cfg0_block()), ("cfg3_vitl14_lift896", lambda: cfg3_block()),
                            ("cfg2_train_vits14_loftup224", lambda: train_block("dinov2_vits14", "loftup", 8, 224)),
                            ("fp32_mode", lambda: fp32_mode_block(args.arch, args.upsampler, B, S))):
                try:
                    line[key] = fn()
                except Exception as exc:
                    line[key] = {"error": repr(exc)}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, S, args.upsampler, vit, seed=1000, quick=args.cpu_baseline_quick)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def run_train(args):
    world, rank, dist, barrier, max_over_ranks = _dist_setup(args)
    import logging
    logging.getLogger("root").setLevel(logging.WARNING)
    from isegprobe_amd.core.training.trainer import DataParallelTrainer

    vit = VITS[args.arch]
    torch.manual_seed(0)
    model = build(args.upsampler, args.size, args.arch).cuda()  # identical replicas: same seed on every rank
    trainer = DataParallelTrainer(model, lr=5e-5)
    batch = synthetic_train_batch(args.batch, args.size, seed=2000 + rank)
    ar_ms = []
    orig = trainer.bucket.finish_overlapped

    def timed_finish(*a, **k):  # HIP events around what is left of the gradient all-reduce behind backward: the head's slice was
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)  # issued inside it (GradBucket.arm_early);
        s.record()                                                                        # here: the click encoder's slice,
        r = orig(*a, **k)                                                                 # the wait for both, the division
        e.record()
        ar_ms.append((s, e))
        return r
    for _ in range(args.warmup):
        trainer.step(batch, num_iters=args.sim_clicks)
    barrier()
    trainer.bucket.finish_overlapped = timed_finish
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(batch, num_iters=args.sim_clicks)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    assert torch.isfinite(loss)
    if rank == 0:
        B, S, D = args.batch, args.size, vit["embed_dim"]
        nbytes = trainer.bucket.nbytes()
        ar = float(np.mean([s.elapsed_time(e) for s, e in ar_ms])) if ar_ms else None
        line = {
            "metric": "images/sec, SBD-shaped train step (fwd + HIP backward + RCCL gradient all-reduce + Adam)",
            "value": world * B * args.steps / dt, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{args.arch} + {args.upsampler} + ConvSegHead({D},2,1), {S}x{S} crops, batch {B}/GPU, clicks before "
                                   f"the backbone, {args.sim_clicks} simulated corrective clicks per step, seeded random-init weights",
                       "per_gpu_batch": B, "global_batch": world * B, "image_size": S, "parallelism": f"dp{world}"},
            "collective": {"op": "all_reduce(sum)/world of ONE flat fp32 bucket (trainable gradients: embed_coords + head), the head's slice issued "
                                 "from inside backward; ms_per_step = the part exposed behind backward (HIP events around finish_overlapped)",
                           "backend": (os.environ.get("ISEGPROBE_DIST_BACKEND") or "nccl (RCCL)") if world > 1 else "none (single process)", "bytes": nbytes,
                           "ms_per_step": ar if world > 1 else 0.0,
                           "bus_GBs": (2.0 * (world - 1) / world * nbytes / (ar * 1e-3) / 1e9) if (world > 1 and ar) else None},
            "peak_mem_GiB": torch.cuda.max_memory_allocated() / 2 ** 30,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world, port, argv):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.argv = argv
    main()


def _spawn(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks here.  The parent never touches the GPU and
    never replaces itself; it starts fresh interpreters (spawn) and waits for them."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, args.gpus, port, list(sys.argv))) for r in range(args.gpus)]
    for p in procs:
        p.start()
    code = 0
    for p in procs:
        p.join()
        code = code or (p.exitcode or 0)
    raise SystemExit(code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--mode", choices=("forward", "train"), default="forward")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (forward: 32; train: 32)")
    ap.add_argument("--size", type=int, default=None, help="forward: 448; train: 224 (reference crop_size, train_cfg.yaml:22)")
    ap.add_argument("--arch", default=None, help="dinov2_vits14 (forward default, train on 1 GPU) | dinov2_vitb14 (train, "
                                                 "configs[4]) | dinov2_vitl14")
    ap.add_argument("--upsampler", default=None, help="forward: jbu_featup; train: loftup")
    ap.add_argument("--sim-clicks", type=int, default=0, help="train: simulated corrective clicks (no-grad forwards) per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="batch-8 CPU baseline from one run instead of the median of 3")
    ap.add_argument("--no-stages", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the loftup448 / size896 / cfg3_vitl14_lift896 blocks (north_star's other named numbers, BASELINE configs[3])")
    ap.add_argument("--no-alt", action="store_true", help="skip the second timed region (bf16 head convolutions, alt_head_bf16)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    train = args.mode == "train"
    args.steps = args.steps if args.steps is not None else (10 if train else 20)
    args.warmup = args.warmup if args.warmup is not None else (3 if train else 5)
    args.batch = args.batch if args.batch is not None else 32
    args.size = args.size if args.size is not None else (224 if train else 448)
    args.arch = args.arch or ("dinov2_vitb14" if train else "dinov2_vits14")
    args.upsampler = args.upsampler or ("loftup" if train else "jbu_featup")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _spawn(args)
    (run_dry if args.dry_run else run_train if train else run_forward)(args)


if __name__ == "__main__":
    main()
