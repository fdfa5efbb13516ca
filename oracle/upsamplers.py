"""Oracle: feature upsamplers (test infrastructure only).

torch-CPU fp32 functional restatements.  Weights are flat dicts with the
reference's state-dict key names (relative to the upsampler module).
"""
import torch
import torch.nn.functional as F


# --- basic (core/model/upsamplers/basic_upsamplers.py:8-42) -----------------
def identity(source, guidance):
    return source


def nearest(source, guidance):
    return F.interpolate(source, guidance.shape[2:], mode="nearest")


def bilinear(source, guidance):
    return F.interpolate(source, guidance.shape[2:], mode="bilinear", align_corners=True)


def bicubic(source, guidance):
    return F.interpolate(source, guidance.shape[2:], mode="bicubic")


def _bn_eval(x, w, prefix, eps=1e-5, train=False, stats=None):
    """nn.BatchNorm2d.  train=False: running statistics (eval mode).  train=True: the reference's net.train()
    (core/training/trainer.py:214,431) also flips the FROZEN upsamplers' BatchNorm2d layers to batch statistics
    (LiFT.py:19-24,71-76,88; loftup/loftup.py:58,63); `stats`, if given, collects the running statistics the
    train-mode forward would leave behind (momentum 0.1, unbiased variance)."""
    if not train:
        return F.batch_norm(x, w[prefix + "running_mean"], w[prefix + "running_var"],
                            w[prefix + "weight"], w[prefix + "bias"], False, 0.0, eps)
    rm, rv = w[prefix + "running_mean"].clone(), w[prefix + "running_var"].clone()
    y = F.batch_norm(x, rm, rv, w[prefix + "weight"], w[prefix + "bias"], True, 0.1, eps)
    if stats is not None:
        stats[prefix + "running_mean"], stats[prefix + "running_var"] = rm, rv
    return y


# --- LiFT (core/model/upsamplers/LiFT.py:47-122), eval-mode BatchNorm --------
def lift(source, guidance, w, prefix="lift.", bn_train=False, stats=None, capture=None):
    """LiFTUpsampler.forward(source, guidance) = LiFT(imgs=guidance, x=source) (:145-146).
    Returns [B, C, 2h, 2w]."""
    import functools
    _bn_eval = functools.partial(globals()["_bn_eval"], train=bn_train, stats=stats)
    g = lambda k: w[prefix + k]
    ws = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}
    x = source
    # image_convs_1: 3->32 s2, BN, ReLU, 32->32 s2, BN, ReLU  (:70-77)
    i1 = F.relu(_bn_eval(F.conv2d(guidance, g("image_convs_1.0.weight"), g("image_convs_1.0.bias"),
                                  stride=2, padding=1), ws, "image_convs_1.1."))
    i1 = F.relu(_bn_eval(F.conv2d(i1, g("image_convs_1.3.weight"), g("image_convs_1.3.bias"),
                                  stride=2, padding=1), ws, "image_convs_1.4."))
    i1 = F.adaptive_max_pool2d(i1, (x.shape[2] * 2, x.shape[3] * 2))  # :110
    # image_convs_2: 32->32 s2, BN, ReLU (:87-91)
    i2 = F.relu(_bn_eval(F.conv2d(i1, g("image_convs_2.0.weight"), g("image_convs_2.0.bias"),
                                  stride=2, padding=1), ws, "image_convs_2.1."))
    x = torch.cat([x, i2], dim=1)  # :117
    # Up (:30-44): ConvTranspose2d(k2,s2) -> cat imgs_1 -> DoubleConv (no conv bias, :18-25)
    x = F.conv_transpose2d(x, g("up1.up.weight"), g("up1.up.bias"), stride=2)
    x = torch.cat([x, i1], dim=1)
    x = F.relu(_bn_eval(F.conv2d(x, g("up1.conv_1.double_conv.0.weight"), None, padding=1),
                        ws, "up1.conv_1.double_conv.1."))
    if capture is not None:  # (tests: the ReLU masks of the two DoubleConv layers, the only non-linearities on the source's path)
        capture["lift_relu1"] = (x > 0).detach()
    x = F.relu(_bn_eval(F.conv2d(x, g("up1.conv_1.double_conv.3.weight"), None, padding=1),
                        ws, "up1.conv_1.double_conv.4."))
    if capture is not None:
        capture["lift_relu2"] = (x > 0).detach()
    return F.conv2d(x, g("outc.weight"), g("outc.bias"))  # :119


# --- LoftUp (core/model/upsamplers/loftup/loftup.py:100-149, layers.py) ------
def _implicit_feats(img, biases, n_freqs, color):
    """ImplicitFeaturizer.forward (loftup/layers.py:107-158) with learn_bias=True."""
    b, _, h, w = img.shape
    gh = torch.linspace(-1, 1, h)
    gw = torch.linspace(-1, 1, w)
    grid = torch.stack(torch.meshgrid(gh, gw, indexing="ij")).unsqueeze(0).expand(b, 2, h, w)
    feats = torch.cat([grid, img], dim=1) if color else grid
    mult = feats.shape[1]
    feats = feats.unsqueeze(1)  # [b,1,mult,h,w]
    freqs = torch.exp(torch.linspace(-2, 10, n_freqs)).reshape(1, n_freqs, 1, 1, 1)
    feats = feats * freqs  # [b,F,mult,h,w]
    s = (feats + biases[0].reshape(1, n_freqs, mult, 1, 1)).reshape(b, n_freqs * mult, h, w)
    c = (feats + biases[1].reshape(1, n_freqs, mult, 1, 1)).reshape(b, n_freqs * mult, h, w)
    out = [torch.sin(s), torch.cos(c)]
    if color:
        out.append(img)
    return torch.cat(out, dim=1)


def _minmax(x):
    """MinMaxScaler (layers.py:61-71): per-channel min/max over the WHOLE batch."""
    c = x.shape[1]
    flat = x.permute(1, 0, 2, 3).reshape(c, -1)
    lo = flat.min(dim=-1).values.reshape(1, c, 1, 1)
    sc = flat.max(dim=-1).values.reshape(1, c, 1, 1) - lo
    return (x - lo) / sc.clamp_min(0.0001) - 0.5


def _channel_ln(x, weight, bias, eps=1e-5):
    """ChannelNorm (layers.py:26-35): nn.LayerNorm over C of an NCHW tensor."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), weight, bias, eps).permute(0, 3, 1, 2)


def _convnext_ln(x, weight, bias, eps=1e-6):
    """LayerNorm variant (layers.py:38-58)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return weight[:, None, None] * x + bias[:, None, None]


def _mha(q, kv, w, prefix, heads):
    """nn.MultiheadAttention forward as used by CrossAttentionLayer (layers.py:186-202):
    packed in_proj, scaled dot-product, out_proj.  q: [B,Lq,E], kv: [B,Lk,E]."""
    E = q.shape[-1]
    W, bvec = w[prefix + "in_proj_weight"], w[prefix + "in_proj_bias"]
    Q = F.linear(q, W[:E], bvec[:E])
    K = F.linear(kv, W[E:2 * E], bvec[E:2 * E])
    V = F.linear(kv, W[2 * E:], bvec[2 * E:])
    B, Lq, _ = Q.shape
    Lk = K.shape[1]
    hd = E // heads
    Q = Q.reshape(B, Lq, heads, hd).transpose(1, 2) * hd ** -0.5
    K = K.reshape(B, Lk, heads, hd).transpose(1, 2)
    V = V.reshape(B, Lk, heads, hd).transpose(1, 2)
    out = torch.empty(B, heads, Lq, hd)
    step = 16384  # chunk the queries: the oracle must not need the 3.3 GB weight matrix
    for s in range(0, Lq, step):
        p = (Q[:, :, s:s + step] @ K.transpose(-2, -1)).softmax(dim=-1)
        out[:, :, s:s + step] = p @ V
    out = out.transpose(1, 2).reshape(B, Lq, E)
    return F.linear(out, w[prefix + "out_proj.weight"], w[prefix + "out_proj.bias"])


def loftup(source, guidance, w, prefix="upsampler.", heads=4, n_freqs=20, depth=2, bn_train=False, stats=None):
    """LoftUpUpsampler.forward -> UpsamplerwithChannelNorm (loftup.py:141-149) -> LoftUp.forward
    (:100-138), lr_pe_type="sine"; BatchNorm in eval mode unless bn_train.  Key prefixes: ``channelnorm.``
    and ``upsampler.`` under ``prefix``."""
    import functools
    _bn_eval = functools.partial(globals()["_bn_eval"], train=bn_train, stats=stats)
    ws = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}
    g = lambda k: ws[k]
    lr = _channel_ln(source, g("channelnorm.norm.weight"), g("channelnorm.norm.bias"))
    u = "upsampler."
    # fourier_feat = MinMaxScaler -> ImplicitFeaturizer(color, 20 freqs, learn_bias) (:48-51)
    x = _implicit_feats(_minmax(guidance), g(u + "fourier_feat.1.biases"), n_freqs, True)
    # first_conv (:53-63): ChannelNorm, conv3x3+BN+ReLU, conv3x3+BN+ReLU
    x = _channel_ln(x, g(u + "first_conv.0.norm.weight"), g(u + "first_conv.0.norm.bias"))
    x = F.relu(_bn_eval(F.conv2d(x, g(u + "first_conv.1.weight"), g(u + "first_conv.1.bias"), padding=1),
                        ws, u + "first_conv.2."))
    x = F.relu(_bn_eval(F.conv2d(x, g(u + "first_conv.4.weight"), g(u + "first_conv.4.bias"), padding=1),
                        ws, u + "first_conv.5."))
    b, c, h, wd = x.shape
    q = x.flatten(2).permute(0, 2, 1)  # (B, HW, c)
    # LR tokens + sine PE (:114-118): ImplicitFeaturizer(color_feats=False, n_freqs=5)
    pe = _implicit_feats(lr, g(u + "lr_pe.biases"), 5, False)
    kv = torch.cat([lr, pe], dim=1).flatten(2).permute(0, 2, 1)
    # CATransformer (layers.py:205-228)
    for i in range(depth):
        p = f"{u}ca_transformer.layers.{i}."
        qn = F.layer_norm(q, (c,), g(p + "0.norm_q.weight"), g(p + "0.norm_q.bias"))
        kn = F.layer_norm(kv, (c,), g(p + "0.norm_kv.weight"), g(p + "0.norm_kv.bias"))
        q = _mha(qn, kn, ws, p + "0.attention.", heads) + q
        f = F.layer_norm(q, (c,), g(p + "1.net.0.weight"), g(p + "1.net.0.bias"))
        f = F.gelu(F.linear(f, g(p + "1.net.1.weight"), g(p + "1.net.1.bias")))
        q = F.linear(f, g(p + "1.net.4.weight"), g(p + "1.net.4.bias")) + q
    q = F.layer_norm(q, (c,), g(u + "ca_transformer.norm.weight"), g(u + "ca_transformer.norm.bias"))
    x = q.permute(0, 2, 1).reshape(b, c, h, wd)
    # final_conv (:65-68): 1x1 conv -> channel LayerNorm (eps 1e-6)
    x = F.conv2d(x, g(u + "final_conv.0.weight"), g(u + "final_conv.0.bias"))
    return _convnext_ln(x, g(u + "final_conv.1.weight"), g(u + "final_conv.1.bias"))


# --- FeatUp JBU stack --------------------------------------------------------
# THIRD-PARTY, ABSENT FROM /root/reference: module mhamilton723/FeatUp (un-pinned VCS
# dependency, reference requirements.txt:28), reached through
# core/model/upsamplers/JBUFeatUp.py:30-32 (`torch.hub.load(...).upsampler`).
# Restated from the published algorithm (featup/upsamplers.py: JBULearnedRange,
# JBUStack; featup/adaptive_conv_cuda: AdaptiveConv).  *Parity unpinned*: the reference
# holds no test or fixture for it and the package cannot be imported here.
def _jbu_stage(source, guidance, w, p, radius=3, drops=None):
    """drops (train mode, the reference's net.train() on the frozen stack): (range [B,32], fixup [B,49]) Dropout2d
    multipliers -- 0 or 1/(1-p) per (image, channel) -- applied where the modules' Dropout2d(0.1) layers sit."""
    d = 2 * radius + 1
    GB, GC, GH, GW = guidance.shape
    # range kernel: 1x1(3->32) GELU Dropout2d 1x1(32->32); 49-tap dot; softmax * temp
    hid = F.gelu(F.conv2d(guidance, w[p + "range_proj.0.weight"], w[p + "range_proj.0.bias"]))
    if drops is not None:
        hid = hid * drops[0][:, :, None, None]
    proj = F.conv2d(hid, w[p + "range_proj.3.weight"], w[p + "range_proj.3.bias"])
    key_dim = proj.shape[1]
    padded = F.pad(proj, [radius] * 4, mode="reflect")
    queries = F.unfold(padded, d).reshape(GB, key_dim, d * d, GH, GW).permute(0, 1, 3, 4, 2)
    temp = w[p + "range_temp"].exp().clamp_min(1e-4).clamp_max(1e4)
    range_k = F.softmax(temp * torch.einsum("bchwp,bchw->bphw", queries, proj), dim=1)
    # spatial Gaussian on linspace(-1,1,d)^2
    dr = torch.linspace(-1, 1, d)
    xx, yy = torch.meshgrid(dr, dr, indexing="ij")
    spatial = torch.exp(-(xx.square() + yy.square()) / (2 * w[p + "sigma_spatial"] ** 2)).reshape(1, d * d, 1, 1)
    k = range_k * spatial
    k = k / k.sum(1, keepdim=True).clamp(1e-7)
    fix = F.gelu(F.conv2d(torch.cat([k, guidance], dim=1), w[p + "fixup_proj.0.weight"], w[p + "fixup_proj.0.bias"]))
    if drops is not None:
        fix = fix * drops[1][:, :, None, None]
    fix = F.conv2d(fix, w[p + "fixup_proj.3.weight"], w[p + "fixup_proj.3.bias"])
    k = k + 0.1 * fix
    k = k.permute(0, 2, 3, 1).reshape(GB, GH, GW, d, d)
    hr = F.interpolate(source, (GH, GW), mode="bicubic", align_corners=False)
    hr = F.pad(hr, [radius] * 4, mode="reflect")
    # AdaptiveConv: out[b,c,y,x] = sum_ij hr[b,c,y+i,x+j] * k[b,y,x,i,j]
    out = torch.zeros(GB, source.shape[1], GH, GW)
    for i in range(d):
        for j in range(d):
            out += hr[:, :, i:i + GH, j:j + GW] * k[:, None, :, :, i, j]
    return out


def jbu_stack(source, guidance, w, prefix="upsampler.", drops=None):
    """drops: None (eval), or the train-mode Dropout2d multipliers {"stages": [(range [B,32], fixup [B,49])] * 4,
    "fixup": [B,C]} (see _jbu_stage)."""
    ws = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)}
    x = source
    for s in range(1, 5):
        h, wd = x.shape[2:]
        small = F.adaptive_avg_pool2d(guidance, (h * 2, wd * 2))
        x = _jbu_stage(x, small, ws, f"up{s}.", drops=None if drops is None else drops["stages"][s - 1])
    # fixup_proj = Sequential(Dropout2d(0.2), Conv2d 1x1): index 1
    xin = x if drops is None else x * drops["fixup"][:, :, None, None]
    return F.conv2d(xin, ws["fixup_proj.1.weight"], ws["fixup_proj.1.bias"]) * 0.1 + x
