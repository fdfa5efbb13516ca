"""fp32-accurate forward of the probed path on the bf16 MFMA engine (the "logits within 1e-3 fp32" gate of
BASELINE.json's north_star; the product path is the bf16 one and is held to 1e-2).

Every contraction (patch embedding, the ViT's linears, QK^T, PV, the head's 3x3 convs and classifier) runs as
"three bf16 products": an fp32 operand is split into hi = bf16(x) and lo = bf16(x - hi); laid out along K as
[hi | hi | lo] against weights [whi | wlo | whi] the existing GEMM / conv kernels accumulate hi.whi + hi.wlo + lo.whi
in fp32 (isp_split_bf16x3; the dropped lo.wlo term is 2^-18 relative).  Everything between the contractions stays
fp32: the residual stream, LayerNorm (fp32 in and out), softmax (isp_softmax_rows_f32), GELU / ReLU (applied by the
split kernel on the way into the next contraction), the align_corners bilinear resizes (isp_resize_bilinear_ac_nchw_f32).
torch is used for layout only (unfold, permute, padding copies).  Three to four times the work of the bf16 path and
unfused: a checking mode, not the fast path.

Scope: DINOv2 featurizer (clicks before / after the backbone, or none), identity / bilinear / LiFT / LoftUp / FeatUp-JBU upsampler,
ConvSegHead / SimpleConvSegHead -- BASELINE.json configs[0], the reference's own CPU-runnable configuration
(models/sbd/dinov2/patch-embed_bilinear.py:40, core/model/iseg_probe_model.py:110-134), and the LiFT / LoftUp probes
(configs[1]-[4]).  FeatUp JBU runs as the published stage-by-stage algorithm in plain fp32 (csrc/jbu_f32.hip).
"""
import torch
import torch.nn.functional as F

from ... import hip_ops as ops
from ..._lib import EP_BIAS_F32, EP_TOKENS_F32, IspError
from .featurizers.DINOv2 import LN_EPS


import os

# ISEGPROBE_FP32_ATTENTION=split: the trunk's attention as per-(batch, head) split-bf16 GEMMs + a softmax pass (rounds 1-3)
_FUSED_ATTENTION = os.environ.get("ISEGPROBE_FP32_ATTENTION", "fused") != "split"


class _WeightSplits:
    """[whi | wlo | whi] images of the parameters, rebuilt when a parameter changes (data pointer / version)."""

    def __init__(self):
        self._c = {}

    def get(self, key, params, build):
        sig = tuple((p.data_ptr(), p._version) for p in params)
        hit = self._c.get(key)
        if hit is None or hit[0] != sig:
            with torch.no_grad():
                hit = (sig, build())
            self._c[key] = hit
        return hit[1]


def _w3(cache, key, weight2d_fn, *params):
    return cache.get(key, params, lambda: ops.split3(weight2d_fn().detach().float().contiguous(), weights=True))


def _linear(x, w3, bias, act=None, scale=1.0):
    """fp32 [M,K] x W^T (+ bias) -> fp32 [M,N]; `act` / `scale` apply to x first."""
    return ops.linear(ops.split3(x, act=act, scale=scale), w3, bias, None, out_dtype=torch.float32)


def _residual(x, h, w3, bias, gamma, act=None):
    ops.linear_residual_(x, ops.split3(h, act=act), w3, bias, gamma)


def _attention(qkv, B, L, heads, scale):
    """softmax(q k^T * scale) v per (batch, head) from the packed fp32 qkv [B*L, 3*heads*64] (attention.py:54-71)."""
    if _FUSED_ATTENTION:  # one launch per block: flash form on the f32-input matrix instruction (csrc/attention_f32.hip)
        return ops.attention_packed_qkv_f32(qkv.contiguous(), B, L, heads, scale)
    D = heads * 64
    Lp = (L + 3) // 4 * 4  # GEMM output columns come in fours: zero key rows, masked out by the softmax
    out = torch.empty(B * L, D, device=qkv.device, dtype=torch.float32)
    kp = torch.zeros(Lp, 64, device=qkv.device, dtype=torch.float32)
    vt = torch.zeros(64, Lp, device=qkv.device, dtype=torch.float32)
    for b in range(B):
        rows = qkv[b * L:(b + 1) * L]
        for h in range(heads):
            q, k, v = (rows[:, i * D + h * 64: i * D + (h + 1) * 64] for i in range(3))
            kp[:L].copy_(k)
            vt[:, :L].copy_(v.t())
            s = ops.linear(ops.split3(q, scale=scale), ops.split3(kp, weights=True), None, None, out_dtype=torch.float32)
            ops.softmax_rows_(s, L)
            ops.gemm(ops.split3(s), ops.split3(vt, weights=True),
                     ops._epilogue(EP_BIAS_F32, out[b * L:(b + 1) * L, h * 64:(h + 1) * 64], D, None))
    return out


def _vit_trunk(fz, cache, x, B, T):
    """The transformer blocks on the fp32 token stream x [B*(T+1), D] (block.py:92-117, attention.py:54-71, mlp.py:34-40;
    modified in place) and the read-out: final LayerNorm with the cls row dropped (DINOv2.py:533-546), or, for the DINO
    ViT-S/16 featurizer with feat_type="key", the keys of the last block (DINO.py:583-590).  -> fp32 [B*T, D]."""
    from .featurizers.DINO import DINOFeaturizer
    m = fz.model
    D, heads = m.embed_dim, m.num_heads
    if D // heads != 64:
        raise IspError("forward_fp32 is built for head_dim 64")
    L = T + 1
    f32 = lambda t: t.detach().float().contiguous()
    last_keys = isinstance(fz, DINOFeaturizer) and fz.feat_type == "key"
    nblk = len(m.blocks)
    for i, blk in enumerate(m.blocks):
        g1 = f32(blk.ls1.gamma) if hasattr(blk.ls1, "gamma") else None
        g2 = f32(blk.ls2.gamma) if hasattr(blk.ls2, "gamma") else None
        y = ops.layernorm(x, f32(blk.norm1.weight), f32(blk.norm1.bias), LN_EPS, out_dtype=torch.float32)
        qkv = _linear(y, _w3(cache, ("qkv", i), lambda: blk.attn.qkv.weight, blk.attn.qkv.weight), f32(blk.attn.qkv.bias))
        if last_keys and i == nblk - 1:
            k = qkv.view(B, L, 3, heads, 64)[:, 1:, 1]                     # [B,T,heads,64], cls removed
            return k.permute(0, 1, 3, 2).reshape(B * T, D).contiguous()    # channel = d*heads + head
        att = _attention(qkv, B, L, heads, 64 ** -0.5)
        _residual(x, att, _w3(cache, ("proj", i), lambda: blk.attn.proj.weight, blk.attn.proj.weight),
                  f32(blk.attn.proj.bias), g1)
        y = ops.layernorm(x, f32(blk.norm2.weight), f32(blk.norm2.bias), LN_EPS, out_dtype=torch.float32)
        hid = _linear(y, _w3(cache, ("fc1", i), lambda: blk.mlp.fc1.weight, blk.mlp.fc1.weight), f32(blk.mlp.fc1.bias))
        _residual(x, hid, _w3(cache, ("fc2", i), lambda: blk.mlp.fc2.weight, blk.mlp.fc2.weight),
                  f32(blk.mlp.fc2.bias), g2, act="gelu")
    return ops.layernorm(x, f32(m.norm.weight), f32(m.norm.bias), LN_EPS, out_dtype=torch.float32,
                         group_out=T, skip=1, rows_out=B * T)  # [B*T, D] = NHWC [B,h,w,D]


def featurizer_fp32(fz, image, additional_features=None):
    """DINOv2Featurizer / DINOFeaturizer .forward(x, additional_features) (DINOv2.py:500-546, DINO.py:529-611) in
    fp32-accurate arithmetic: image [B,3,H,W] (already normalised), click tokens [B,T,D] or None -> [B,D,h,w] fp32."""
    cache = fz.__dict__.setdefault("_fp32_splits", _WeightSplits())
    with torch.no_grad():
        m = fz.model
        p = fz.patch_size
        image = image.float().contiguous()
        B, _, H, W = image.shape
        h, w = H // p, W // p
        T, D = h * w, m.embed_dim
        mode = fz.feats_injection_mode
        inject = additional_features is not None and mode != "no_injection"
        A = _patch_matrix(fz, image, None)
        pw, pb = m.patch_embed.proj.weight, m.patch_embed.proj.bias
        table, cls_row = fz._pos_embed(H, W)
        x = torch.empty(B * (T + 1), D, device=image.device, dtype=torch.float32)
        ops.gemm(ops.split3(A), _w3(cache, "embed_img", lambda: pw.flatten(1), pw),
                 ops._epilogue(EP_TOKENS_F32, x, D, pb.detach().float().contiguous(), None, table, T))
        x.view(B, T + 1, D)[:, 0].copy_(cls_row)
        if inject and mode == "before_backbone":
            x.view(B, T + 1, D)[:, 1:] += additional_features.float()
        feats = _vit_trunk(fz, cache, x, B, T)
        if inject and mode == "after_backbone":
            feats = feats + additional_features.float().reshape(B * T, D)
        return feats.view(B, h, w, D).permute(0, 3, 1, 2)


def maskclip_featurizer_fp32(fz, image, additional_features=None):
    """MaskCLIPFeaturizer.forward (MaskCLIP.py:41-92, maskclip/model.py:251-263,321-430) in fp32-accurate arithmetic:
    CLIP ViT with ln_pre, QuickGELU blocks, the last block's VALUE path only (no attention, no residual), ln_post and
    the output projection -> [B, output_dim, h, w] fp32.  (The reference casts CLIP to fp16 on CUDA; the fixtures and
    this mode are the fp32 computation.)"""
    cache = fz.__dict__.setdefault("_fp32_splits", _WeightSplits())
    with torch.no_grad():
        v = fz.model.visual
        p = fz.patch_size
        image = image.float().contiguous()
        B, _, H, W = image.shape
        h, w = H // p, W // p
        T, D, heads = h * w, v.width, v.heads
        mode = fz.feats_injection_mode
        before = additional_features is not None and mode == "before_backbone"
        after = additional_features is not None and mode == "after_backbone"
        f32 = lambda t: t.detach().float().contiguous()
        table, cls_row = fz._pos(w, h, H, W) if before else fz._pos(h, w, H, W)  # (the reference's swapped grid, kept)
        A = F.unfold(image, p, stride=p).transpose(1, 2).reshape(B * T, -1).contiguous()
        x = torch.empty(B * (T + 1), D, device=image.device, dtype=torch.float32)
        ops.gemm(ops.split3(A), _w3(cache, "conv1", lambda: v.conv1.weight.flatten(1), v.conv1.weight),
                 ops._epilogue(EP_TOKENS_F32, x, D, torch.zeros(D, device=image.device), None, table, T))
        x.view(B, T + 1, D)[:, 0].copy_(cls_row)
        if before:
            x.view(B, T + 1, D)[:, 1:] += additional_features.float()
        x = ops.layernorm(x, f32(v.ln_pre.weight), f32(v.ln_pre.bias), 1e-5, out_dtype=torch.float32)
        blocks = list(v.transformer.resblocks)
        for i, blk in enumerate(blocks):
            a = ops.layernorm(x, f32(blk.ln_1.weight), f32(blk.ln_1.bias), 1e-5, out_dtype=torch.float32)
            ipw, ipb = blk.attn.in_proj_weight, blk.attn.in_proj_bias
            wo = _w3(cache, ("out", i), lambda: blk.attn.out_proj.weight, blk.attn.out_proj.weight)
            if i == len(blocks) - 1:  # forward_v: value projection -> out projection, nothing else
                vin = _linear(a, _w3(cache, ("v", i), lambda: ipw[-D:], ipw), f32(ipb[-D:]))
                vout = _linear(vin, wo, f32(blk.attn.out_proj.bias))
                break
            qkv = _linear(a, _w3(cache, ("qkv", i), lambda: ipw, ipw), f32(ipb))
            att = _attention(qkv, B, T + 1, heads, 64 ** -0.5)
            _residual(x, att, wo, f32(blk.attn.out_proj.bias), None)
            m = ops.layernorm(x, f32(blk.ln_2.weight), f32(blk.ln_2.bias), 1e-5, out_dtype=torch.float32)
            m = _linear(m, _w3(cache, ("fc", i), lambda: blk.mlp.c_fc.weight, blk.mlp.c_fc.weight), f32(blk.mlp.c_fc.bias))
            _residual(x, m, _w3(cache, ("pj", i), lambda: blk.mlp.c_proj.weight, blk.mlp.c_proj.weight),
                      f32(blk.mlp.c_proj.bias), None, act="quick_gelu")
        post = ops.layernorm(vout, f32(v.ln_post.weight), f32(v.ln_post.bias), 1e-5, out_dtype=torch.float32,
                             group_out=T, skip=1, rows_out=B * T)
        nout = v.output_dim
        npad = (nout + 3) // 4 * 4
        feats = _linear(post, _w3(cache, "proj", lambda: F.pad(v.proj.t(), (0, 0, 0, npad - nout)), v.proj), None)[:, :nout]
        if after:
            feats = feats + additional_features.float().reshape(B * T, nout)
        return feats.reshape(B, h, w, nout).permute(0, 3, 1, 2)


def _patch_matrix(featurizer, image, coord):
    """im2col of the image (and, before-backbone injection, of [prev_mask | click maps]) for the patch embedding(s) as
    ONE GEMM over the concatenated K axis (DINOv2.py:518-523): fp32 [B*T, (3 + 3) * p * p].  unfold is layout only."""
    p = featurizer.patch_size
    B, _, H, W = image.shape
    if H % p or W % p:
        raise AssertionError(f"Input image size {H}x{W} is not a multiple of patch size {p}")
    cols = [F.unfold(image, p, stride=p)]
    if coord is not None:
        cols.append(F.unfold(coord, p, stride=p))
    return torch.cat(cols, dim=1).transpose(1, 2).reshape(B * (H // p) * (W // p), -1).contiguous()


def _conv3x3(cache, key, y, weight4d, bias, act, params):
    """3x3 / pad 1 conv on fp32 NHWC `y` with weight [N, C, 3, 3] (already BatchNorm-folded where applicable); `act`
    applies to the input on the way in.  -> fp32 NHWC [B, H, W, N]."""
    B, H, W, C = y.shape
    N = weight4d.shape[0]
    w3 = cache.get(key, params, lambda: ops.split3(weight4d.detach().float().permute(0, 2, 3, 1).reshape(N * 9, C).contiguous(),
                                                   weights=True)).view(N, -1)
    a3 = ops.split3(y.reshape(-1, C), act=act).view(B, H, W, -1)
    return ops.conv3x3(a3, w3, bias.detach().float().contiguous(), None, out_dtype=torch.float32)


def _conv3x3_s2(cache, key, x_nchw, weight4d, bias, params):
    """3x3 / stride 2 / pad 1 conv on fp32 NCHW planes as im2col (unfold: layout only) + one split GEMM -> fp32 NCHW."""
    B, C, H, W = x_nchw.shape
    N = weight4d.shape[0]
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    A = F.unfold(x_nchw, 3, padding=1, stride=2).transpose(1, 2).reshape(B * Ho * Wo, C * 9).contiguous()
    w3 = cache.get(key, params, lambda: ops.split3(weight4d.detach().float().flatten(1).contiguous(), weights=True))
    y = ops.linear(ops.split3(A), w3, bias.detach().float().contiguous(), None, out_dtype=torch.float32)
    return y.view(B, Ho, Wo, N).permute(0, 3, 1, 2).contiguous()


def _lift(up, cache, feats_nhwc, image):
    """LiFT(imgs=image, x=features) (reference LiFT.py:106-122) in fp32-accurate arithmetic -> fp32 NHWC [B,2h,2w,C].
    Eval-mode BatchNorm is folded into the conv weights (as on the bf16 path); ReLU on a materialised tensor and the
    adaptive max pool (a selection, exact in any precision) go through torch."""
    from .upsamplers.LiFT import _fold
    L = up.lift
    B, h, w, C = feats_nhwc.shape
    bn_params = lambda conv, bn: (conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    def folded(name, conv, bn):
        return cache.get(("lift_fold", name), bn_params(conv, bn), lambda: _fold(conv, bn))
    g = image.float().contiguous()
    w1, b1 = folded("ic1a", L.image_convs_1[0], L.image_convs_1[1])
    a = torch.relu_(_conv3x3_s2(cache, ("lift", "ic1a"), g, w1, b1, bn_params(L.image_convs_1[0], L.image_convs_1[1])))
    w2, b2 = folded("ic1b", L.image_convs_1[3], L.image_convs_1[4])
    a = torch.relu_(_conv3x3_s2(cache, ("lift", "ic1b"), a, w2, b2, bn_params(L.image_convs_1[3], L.image_convs_1[4])))
    i1 = F.adaptive_max_pool2d(a, (2 * h, 2 * w))                                   # [B,32,2h,2w]
    w3_, b3 = folded("ic2", L.image_convs_2[0], L.image_convs_2[1])
    i2 = torch.relu_(_conv3x3_s2(cache, ("lift", "ic2"), i1, w3_, b3, bn_params(L.image_convs_2[0], L.image_convs_2[1])))
    xin = torch.cat((feats_nhwc, i2.permute(0, 2, 3, 1)), dim=3).reshape(B * h * w, C + 32).contiguous()
    upw, upb = L.up1.up.weight, L.up1.up.bias                                        # ConvTranspose2d [cin, n, 2, 2]
    n = upw.shape[1]
    wup3 = cache.get(("lift", "up"), (upw,), lambda: ops.split3(
        upw.detach().float().permute(2, 3, 1, 0).reshape(4 * n, C + 32).contiguous(), weights=True))
    t = ops.linear(ops.split3(xin), wup3, upb.detach().float().repeat(4).contiguous(), None, out_dtype=torch.float32)
    t = t.view(B, h, w, 2, 2, n).permute(0, 1, 3, 2, 4, 5).reshape(B, 2 * h, 2 * w, n)   # the four taps -> pixel shuffle
    cat = torch.cat((t, i1.permute(0, 2, 3, 1)), dim=3).contiguous()
    dc = L.up1.conv_1.double_conv
    wa, ba = folded("dc1", dc[0], dc[1])
    y = _conv3x3(cache, ("lift", "dc1"), cat, wa, ba, None, bn_params(dc[0], dc[1]))
    wb, bb = folded("dc2", dc[3], dc[4])
    y = _conv3x3(cache, ("lift", "dc2"), y, wb, bb, "relu", bn_params(dc[3], dc[4]))
    wo3 = _w3(cache, ("lift", "outc"), lambda: L.outc.weight.flatten(1), L.outc.weight)
    out = _linear(y.reshape(-1, y.shape[3]), wo3, L.outc.bias.detach().float().contiguous(), act="relu")
    return out.view(B, 2 * h, 2 * w, C)


def _cross_attention(q, k, v, B, Lq, Lk, heads, hd, scale):
    """nn.MultiheadAttention's core (loftup/layers.py:186-202): per (batch, head) softmax(q k^T * scale) v on fp32
    [B*Lq, heads*hd] queries and [B*Lk, heads*hd] keys / values (hd need not be a multiple of anything)."""
    dev = q.device
    Lkp, hdp = (Lk + 3) // 4 * 4, (hd + 3) // 4 * 4
    out = torch.empty(B * Lq, heads * hd, device=dev, dtype=torch.float32)
    kp = torch.zeros(Lkp, hd, device=dev, dtype=torch.float32)
    vt = torch.zeros(hdp, Lkp, device=dev, dtype=torch.float32)
    o = torch.empty(Lq, hdp, device=dev, dtype=torch.float32)
    for b in range(B):
        for h in range(heads):
            sl = slice(h * hd, (h + 1) * hd)
            kp[:Lk].copy_(k[b * Lk:(b + 1) * Lk, sl])
            vt[:hd, :Lk].copy_(v[b * Lk:(b + 1) * Lk, sl].t())
            s = ops.linear(ops.split3(q[b * Lq:(b + 1) * Lq, sl], scale=scale), ops.split3(kp, weights=True), None, None,
                           out_dtype=torch.float32)
            ops.softmax_rows_(s, Lk)
            ops.gemm(ops.split3(s), ops.split3(vt, weights=True), ops._epilogue(EP_BIAS_F32, o, hdp, None))
            out[b * Lq:(b + 1) * Lq, sl].copy_(o[:, :hd])
    return out


def _loftup(up, cache, feats_nhwc, image):
    """LoftUp.forward + the channel-norm wrapper (reference loftup/loftup.py:100-149, layers.py) in fp32-accurate
    arithmetic -> fp32 NHWC [B,H,W,C].  Eval BatchNorm folded; ReLU on the materialised query map goes through torch."""
    lu, cn = up.upsampler.upsampler, up.upsampler.channelnorm
    B, h, w, C = feats_nhwc.shape
    g = image.float().contiguous()
    H, W = g.shape[2:]
    heads = lu.num_heads
    c = C + lu.lr_pe_dim
    hd = c // heads
    M, T = B * H * W, h * w
    f32 = lambda t: t.detach().float().contiguous()
    dev = g.device
    # LR tokens: ChannelNorm(source) ++ sine PE of the LR grid (a table of the learnt biases and (h, w))
    kv = torch.empty(B, T, c, device=dev, dtype=torch.float32)
    kv[:, :, :C] = ops.layernorm(feats_nhwc.reshape(-1, C), f32(cn.norm.weight), f32(cn.norm.bias), cn.norm.eps,
                                 out_dtype=torch.float32).view(B, T, C)
    gh, gw = torch.linspace(-1, 1, h, device=dev), torch.linspace(-1, 1, w, device=dev)
    grid = torch.stack(torch.meshgrid(gh, gw, indexing="ij"))
    fr = torch.exp(torch.linspace(-2, 10, 5, device=dev)).reshape(5, 1, 1, 1)
    pb = lu.lr_pe.biases.detach().float()
    pe = torch.cat([torch.sin(grid.unsqueeze(0) * fr + pb[0].reshape(5, 2, 1, 1)).reshape(10, h, w),
                    torch.cos(grid.unsqueeze(0) * fr + pb[1].reshape(5, 2, 1, 1)).reshape(10, h, w)], 0)
    kv[:, :, C:] = pe.permute(1, 2, 0).reshape(T, 20)
    kv = kv.view(B * T, c)
    # queries: Fourier features -> ChannelNorm -> 2 x (conv3x3 + folded BN + ReLU)
    fin = 10 * lu.n_freqs + 3
    ff = lu.fourier_feat[1]
    fc = lu.first_conv
    x = ops.loftup_fourier_cn(g, ops.minmax_nchw(g), torch.exp(torch.linspace(-2, 10, lu.n_freqs)).to(dev),
                              f32(ff.biases[0].reshape(-1)), f32(ff.biases[1].reshape(-1)), f32(fc[0].norm.weight),
                              f32(fc[0].norm.bias), fin, fc[0].norm.eps, out_dtype=torch.float32)

    def fold(conv, bn):
        sc = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
        return (conv.weight.detach().float() * sc[:, None, None, None],
                (conv.bias.detach().float() - bn.running_mean.float()) * sc + bn.bias.detach().float())
    bnp = lambda conv, bn: (conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var)
    w1, b1 = cache.get(("loftup_fold", 1), bnp(fc[1], fc[2]), lambda: fold(fc[1], fc[2]))
    w2, b2 = cache.get(("loftup_fold", 2), bnp(fc[4], fc[5]), lambda: fold(fc[4], fc[5]))
    x = _conv3x3(cache, ("loftup", "conv1"), x, w1, b1, None, bnp(fc[1], fc[2]))
    x = _conv3x3(cache, ("loftup", "conv2"), x, w2, b2, "relu", bnp(fc[4], fc[5]))
    x = torch.relu_(x).view(M, c)
    scale = hd ** -0.5
    for li, (ca, ffn) in enumerate(lu.ca_transformer.layers):
        ipw, ipb = ca.attention.in_proj_weight, ca.attention.in_proj_bias
        qn = ops.layernorm(x, f32(ca.norm_q.weight), f32(ca.norm_q.bias), ca.norm_q.eps, out_dtype=torch.float32)
        kn = ops.layernorm(kv, f32(ca.norm_kv.weight), f32(ca.norm_kv.bias), ca.norm_kv.eps, out_dtype=torch.float32)
        wq, wk, wv = (_w3(cache, ("loftup", li, n), lambda i=i: ipw[i * c:(i + 1) * c], ipw) for i, n in enumerate("qkv"))
        q = _linear(qn, wq, f32(ipb[:c]))
        k = _linear(kn, wk, f32(ipb[c:2 * c]))
        v = _linear(kn, wv, f32(ipb[2 * c:]))
        a = _cross_attention(q, k, v, B, H * W, T, heads, hd, scale)
        op = ca.attention.out_proj
        _residual(x, a, _w3(cache, ("loftup", li, "o"), lambda: op.weight, op.weight), f32(op.bias), None)  # x += attn
        n0, l1, l2 = ffn.net[0], ffn.net[1], ffn.net[4]
        f = ops.layernorm(x, f32(n0.weight), f32(n0.bias), n0.eps, out_dtype=torch.float32)
        f = _linear(f, _w3(cache, ("loftup", li, "ff1"), lambda: l1.weight, l1.weight), f32(l1.bias))
        _residual(x, f, _w3(cache, ("loftup", li, "ff2"), lambda: l2.weight, l2.weight), f32(l2.bias), None, act="gelu")
    nrm = lu.ca_transformer.norm
    xn = ops.layernorm(x, f32(nrm.weight), f32(nrm.bias), nrm.eps, out_dtype=torch.float32)
    fcv, fln = lu.final_conv[0], lu.final_conv[1]
    y = _linear(xn, _w3(cache, ("loftup", "fin"), lambda: fcv.weight.flatten(1), fcv.weight), f32(fcv.bias))
    out = ops.layernorm(y, f32(fln.weight), f32(fln.bias), fln.eps, out_dtype=torch.float32)
    return out.view(B, H, W, C)


def _jbu(up, cache, feats_nhwc, image):
    """JBUStack.forward (FeatUp, called at JBUFeatUp.py:30-32 of the reference): four x2 stages on the image pooled to
    each stage's size, then x + 0.1 * conv1x1(x) -- in plain fp32, stage by stage as the published algorithm states it
    (csrc/jbu_f32.hip), not through the composite-kernel formulation of the product path."""
    stack = up.upsampler
    g = image.float().contiguous()
    x = feats_nhwc.contiguous()
    f32 = lambda t: t.detach().float().contiguous()
    for st in (stack.up1, stack.up2, stack.up3, stack.up4):
        B, h, w, C = x.shape
        small = ops.adaptive_avg_pool(g, 2 * h, 2 * w)
        proj = ops.jbu_range_proj(small, f32(st.range_proj[0].weight.flatten(1)), f32(st.range_proj[0].bias),
                                  f32(st.range_proj[3].weight.flatten(1)), f32(st.range_proj[3].bias), exact=True)
        x = ops.jbu_stage_f32(x, proj, small, f32(st.fixup_proj[0].weight.flatten(1)), f32(st.fixup_proj[0].bias),
                              f32(st.fixup_proj[3].weight.flatten(1)), f32(st.fixup_proj[3].bias),
                              float(st.range_temp.item()), float(st.sigma_spatial.item()))
    conv = stack.fixup_proj[1]
    B, H, W, C = x.shape
    y = x.reshape(-1, C).clone()
    ops.linear_residual_(y, ops.split3(x.reshape(-1, C)), _w3(cache, ("jbu", "fix"), lambda: conv.weight.flatten(1), conv.weight),
                         f32(conv.bias), torch.full((C,), 0.1, device=x.device, dtype=torch.float32))
    return y.view(B, H, W, C)


def forward_fp32(model, image, points):
    """iSegProbeModel.forward (iseg_base_model.py:67-89 + iseg_probe_model.py:110-134) with fp32-accurate arithmetic."""
    from .featurizers import DINOv2Featurizer
    from .heads.conv_heads import _StackedHead
    from .upsamplers.basic_upsamplers import BilinearUpsampler, IdentityUpsampler
    from .upsamplers.LiFT import LiFTUpsampler
    from .upsamplers.LoftUp import LoftUpUpsampler
    from .upsamplers.JBUFeatUp import JBUFeatUpUpsampler
    fz, head, up = model.backbone, model.head, model.upsampler
    if not isinstance(fz, DINOv2Featurizer) or fz.feats_injection_mode not in ("before_backbone", "after_backbone",
                                                                                "no_injection"):
        raise IspError("forward_fp32 covers the DINOv2 featurizer (clicks before / after the backbone, or none)")
    if not isinstance(up, (BilinearUpsampler, IdentityUpsampler, LiFTUpsampler, LoftUpUpsampler, JBUFeatUpUpsampler)) or \
            not isinstance(head, _StackedHead):
        raise IspError("forward_fp32 covers the identity / bilinear / LiFT / LoftUp / FeatUp-JBU upsamplers and the stacked conv heads")
    cache = model.__dict__.setdefault("_fp32_splits", _WeightSplits())
    save = getattr(model, "_fp32_save", None)  # (set by click_head_gradients_fp32 for the duration of its forward)
    with torch.no_grad():
        image, prev_mask = model.prepare_input(image)
        coord = coord_after = None
        if fz.feats_injection_mode == "before_backbone":
            coord = model.maps_transform(model.get_coord_features(image, prev_mask, points))
        elif fz.feats_injection_mode == "after_backbone":  # DINOv2.py:509-516: click tokens added to the final features
            coord_after = model.maps_transform(model.get_coord_features(image, prev_mask, points))
        m = fz.model
        A = _patch_matrix(fz, image, coord)
        B, _, H, W = image.shape
        h, w = H // fz.patch_size, W // fz.patch_size
        T, D, heads = h * w, m.embed_dim, m.num_heads
        L = T + 1
        pw, pb = m.patch_embed.proj.weight, m.patch_embed.proj.bias
        if coord is not None:
            cw, cb = model.embed_coords.proj.weight, model.embed_coords.proj.bias
            w3 = _w3(cache, "embed", lambda: torch.cat((pw.flatten(1), cw.flatten(1)), dim=1), pw, cw)
            bias = (pb.detach().float() + cb.detach().float()).contiguous()
        else:
            w3 = _w3(cache, "embed", lambda: pw.flatten(1), pw)
            bias = pb.detach().float().contiguous()
        table, cls_row = fz._pos_embed(H, W)
        x = torch.empty(B * L, D, device=image.device, dtype=torch.float32)
        ops.gemm(ops.split3(A), w3, ops._epilogue(EP_TOKENS_F32, x, D, bias, None, table, T))
        x.view(B, L, D)[:, 0].copy_(cls_row)
        f32 = lambda t: t.detach().float().contiguous()
        feats = _vit_trunk(fz, cache, x, B, T)
        if coord_after is not None:
            p_ = fz.patch_size
            Ac = F.unfold(coord_after, p_, stride=p_).transpose(1, 2).reshape(B * T, -1).contiguous()
            cw, cb = model.embed_coords.proj.weight, model.embed_coords.proj.bias
            feats = feats + _linear(Ac, _w3(cache, "embed_after", lambda: cw.flatten(1), cw), f32(cb))
            if save is not None:
                save["Ac"] = Ac
        y = feats.view(B, h, w, D)
        if isinstance(up, LiFTUpsampler):
            y = _lift(up, cache, y, image)
        elif isinstance(up, LoftUpUpsampler):
            y = _loftup(up, cache, y, image)
        elif isinstance(up, JBUFeatUpUpsampler):
            y = _jbu(up, cache, y, image)
        first_done = False
        if not isinstance(up, IdentityUpsampler) and tuple(y.shape[1:3]) != (H, W):
            from .heads.conv_heads import CONV_OF_BILINEAR
            n0 = head.convs[0].conv.out_channels if len(head.convs) else 0
            if (save is None and CONV_OF_BILINEAR and len(head.convs) and head.kernel_size == 3 and y.shape[3] == head.convs[0].conv.in_channels
                    and ops.conv3x3_of_bilinear_supported(y.shape[1], y.shape[2], H, W, n0, torch.float32)):
                # resize + first 3x3 convolution as ONE operator, as on the 16-bit path (heads/conv_heads.py::forward_of_bilinear):
                # the nine tap planes Z_t = y W_t^T at low resolution (three bf16 products, fp32 out), then their bilinear
                # blend in fp32 (csrc/conv_bilinear.hip, exact fp32 arithmetic) -- the [B,H,W,C] fp32 map is never written
                cw0 = head.convs[0].conv.weight
                wz3 = _w3(cache, ("conv_of_bilinear", 0), lambda: cw0.permute(2, 3, 0, 1).reshape(9 * n0, cw0.shape[1]), cw0)
                Bl, hl, wl, Cl = y.shape
                z = _linear(y.reshape(-1, Cl), wz3, None)
                y = ops.conv3x3_of_bilinear_blend(z, f32(head.convs[0].conv.bias), Bl, hl, wl, H, W, n0, relu=False, out_dtype=torch.float32)
                first_done = True
            else:
                # BilinearUpsampler.forward (basic_upsamplers.py:28-33) / the model's resize of a learned upsampler's
                # output to the image size (iseg_probe_model.py:120-129): bilinear, align_corners=True, on fp32 planes
                planes = ops.resize_bilinear_nchw_f32(y.permute(0, 3, 1, 2).contiguous(), H, W)
                y = planes.permute(0, 2, 3, 1).contiguous()
        Bh, Hh, Wh, C = y.shape
        act = "relu" if first_done else None
        if save is not None:
            save.update(geom=(B, h, w, H, W), layer_in=[], lowres=(Hh, Wh) != (h, w))
        for j, layer in enumerate(head.convs):
            if first_done and j == 0:
                continue
            if save is not None:
                save["layer_in"].append((y, act))  # pre-activation input of layer j and the activation applied on the way in
            cw_ = layer.conv.weight
            N = cw_.shape[0]
            k = head.kernel_size
            if k == 3:
                w3 = _w3(cache, ("conv", j), lambda: cw_.permute(0, 2, 3, 1).reshape(N * 9, C), cw_).view(N, -1)
                a3 = ops.split3(y.reshape(-1, C), act=act).view(Bh, Hh, Wh, -1)
                y = ops.conv3x3(a3, w3, f32(layer.conv.bias), None, out_dtype=torch.float32)
            else:  # SimpleConvSegHead: 1x1 convs
                w3 = _w3(cache, ("conv", j), lambda: cw_.flatten(1), cw_)
                y = _linear(y.reshape(-1, C), w3, f32(layer.conv.bias), act=act).view(Bh, Hh, Wh, N)
            C, act = N, "relu"
        if save is not None:
            save["cls_in"] = (y, act)
        clw, clb = head.classifier.weight, head.classifier.bias
        ncls = clw.shape[0]
        npad = (ncls + 3) // 4 * 4
        wc3 = _w3(cache, "cls", lambda: F.pad(clw.flatten(1), (0, 0, 0, npad - ncls)), clw)
        bias = F.pad(clb.detach().float(), (0, npad - ncls)).contiguous()
        logits = _linear(y.reshape(-1, C), wc3, bias, act=act)[:, :ncls]
        logits = logits.reshape(Bh, Hh, Wh, ncls).permute(0, 3, 1, 2).contiguous()
        logits = model._to_image_size(logits, (H, W))
    return {"instances": logits, "instances_aux": None}


def _mm(a, b):
    """fp32 [M, K] x fp32 [N, K]^T -> fp32 [M, N] as three bf16 products (N is padded to a multiple of 4 and cut back)."""
    n = b.shape[0]
    npad = (n + 3) // 4 * 4
    if npad != n:
        b = F.pad(b, (0, 0, 0, npad - n))
    out = ops.linear(ops.split3(a.contiguous()), ops.split3(b.contiguous(), weights=True), None, None, out_dtype=torch.float32)
    return out[:, :n]


def click_head_gradients_fp32(model, image, points, grad_logits, relu_masks=None):
    """Gradients of  sum(logits * grad_logits)  w.r.t. every trainable tensor of a clicks-after-the-backbone model --
    ``embed_coords.proj.{weight,bias}``, ``head.convs.{j}.conv.{weight,bias}``, ``head.classifier.{weight,bias}`` -- in
    fp32-accurate arithmetic: what autograd does for iSegProbeModel.forward (iseg_probe_model.py:110-134,
    heads/conv_heads.py:48-73, featurizers/DINOv2.py:509-516 -- the after_backbone branch, where the frozen trunk is not on
    the gradient's path), with every contraction (the 3x3 convolutions' weight and data gradients, the classifier, the
    click patch-embedding) as "three bf16 products" on the MFMA GEMM (``_mm``) and everything else -- ReLU masks, bias
    sums, the adjoint of the align_corners resize -- in fp32.  The product path's backward runs on bf16 operands and is
    held to cos > 0.99 against autograd of the oracle (a bf16 ReLU-mask flip moves a weight gradient by whole terms); this
    mode separates that rounding noise from an indexing error: it is held to 1e-3 relative
    (tests/test_training_gpu.py::test_gradients_fp32_mode_vs_oracle).  A checking mode: torch does the layout work
    (unfold, transposes), nothing here is fast.  Upsampler: identity or bilinear; head: ConvSegHead.
    ``relu_masks`` (optional, one bool NHWC tensor per conv layer): use these ReLU masks instead of this forward's own --
    a pre-activation within fp32 rounding of zero may fall on either side in two correct fp32 implementations, and one
    flipped entry moves a bias gradient by a whole term (1e-2 relative on 6 272-pixel sums); the test imposes the
    oracle's masks and separately bounds how many entries differ.  Returns (grads, this forward's own masks)."""
    from .heads.conv_heads import ConvSegHead
    from .upsamplers.basic_upsamplers import BilinearUpsampler, IdentityUpsampler
    fz, up, head = model.backbone, model.upsampler, model.head
    if getattr(fz, "feats_injection_mode", None) != "after_backbone" or not isinstance(up, (BilinearUpsampler, IdentityUpsampler)) \
            or not isinstance(head, ConvSegHead) or head.num_classes != 1:
        raise IspError("click_head_gradients_fp32 covers clicks after the backbone, identity / bilinear upsampling and ConvSegHead")
    save = {}
    model._fp32_save = save
    try:
        logits = forward_fp32(model, image, points)["instances"]
    finally:
        del model._fp32_save
    B, h, w, H, W = save["geom"]
    grads = {}
    with torch.no_grad():
        y_last, act_last = save["cls_in"]
        Hh, Wh = y_last.shape[1:3]
        gl = grad_logits.float().contiguous()
        if tuple(logits.shape[2:]) != (Hh, Wh):  # identity upsampler: the logits were resized to the image size
            gl = ops.resize_bilinear_nchw_f32_bwd(gl, Hh, Wh)                         # (iseg_base_model.py:75-80, adjoint)
        M = B * Hh * Wh
        g = gl.permute(0, 2, 3, 1).reshape(M, 1).contiguous()
        a = y_last.reshape(M, -1)
        a = torch.relu(a) if act_last == "relu" else a
        clw = head.classifier.weight.detach().float().flatten(1)                      # [1, C]
        grads["head.classifier.weight"] = _mm(g.t().contiguous(), a.t().contiguous()).view_as(head.classifier.weight)
        grads["head.classifier.bias"] = g.sum(0)
        g_a = g * clw                                                                  # [M, C]: rank one, no contraction
        used_masks = [None] * len(head.convs)
        for j in reversed(range(len(head.convs))):
            y_out = save["layer_in"][j + 1][0] if j + 1 < len(head.convs) else y_last  # this layer's pre-activation output
            mask = (y_out > 0) if relu_masks is None else relu_masks[j].to(y_out.device)
            used_masks[j] = y_out > 0
            g_z = (g_a * mask.reshape(M, -1)).contiguous()                             # ReLU behind every layer (ConvModule)
            y_in, act_in = save["layer_in"][j]
            x_in = torch.relu(y_in) if act_in == "relu" else y_in                      # [B, Hh, Wh, Cin] fp32
            Cin, N = x_in.shape[3], g_z.shape[1]
            conv = head.convs[j].conv
            U = F.unfold(x_in.permute(0, 3, 1, 2), 3, padding=1).transpose(1, 2).reshape(M, Cin * 9)   # (c, ky, kx) order
            grads[f"head.convs.{j}.conv.weight"] = _mm(g_z.t().contiguous(), U.t().contiguous()).view_as(conv.weight)
            grads[f"head.convs.{j}.conv.bias"] = g_z.sum(0)
            # data gradient: g_x[m, c] = sum_{n, tap} g_z[m - tap, n] w[n, c, tap]  = unfold(g_z) . rot180(w)
            Ug = F.unfold(g_z.view(B, Hh, Wh, N).permute(0, 3, 1, 2), 3, padding=1).transpose(1, 2).reshape(M, N * 9)
            wr = conv.weight.detach().float().flip(2, 3).permute(1, 0, 2, 3).reshape(Cin, N * 9)  # [c, (n, ky, kx)]
            g_a = _mm(Ug, wr)
        g_map = g_a.view(B, Hh, Wh, -1).permute(0, 3, 1, 2).contiguous()             # gradient of the head's input, NCHW
        if (Hh, Wh) != (h, w):  # adjoint of the bilinear (align_corners) upsampling to the image size
            g_map = ops.resize_bilinear_nchw_f32_bwd(g_map, h, w)
        g_tok = g_map.permute(0, 2, 3, 1).reshape(B * h * w, -1).contiguous()          # [B*T, D]: d features = d click tokens
        cw = model.embed_coords.proj.weight
        grads["embed_coords.proj.weight"] = _mm(g_tok.t().contiguous(), save["Ac"].t().contiguous()).view_as(cw)
        grads["embed_coords.proj.bias"] = g_tok.sum(0)
    return grads, used_masks
