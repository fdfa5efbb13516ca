"""Oracle: DINOv2 ViT featurizer with click injection (test infrastructure only).

Functional torch-CPU fp32 restatement.  Weights come in as a flat dict using
the reference's DINOv2 hub key layout (``cls_token, pos_embed, patch_embed.proj.*,
blocks.{i}.{norm1,attn.qkv,attn.proj,ls1.gamma,norm2,mlp.fc1,mlp.fc2,ls2.gamma}.*,
norm.*``; reference core/model/featurizers/DINOv2.py:53-180).
"""
import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-6  # DINOv2.py:98  partial(nn.LayerNorm, eps=1e-6)


def patch_tokens(x, weight, bias, patch):
    """Conv k=p,s=p then BCHW -> B(hw)C (dinov2/layers/patch_embed.py:71-87 and
    featurizers/utils/patch_embed.py:37-42 do the same thing)."""
    assert x.shape[2] % patch == 0 and x.shape[3] % patch == 0  # patch_embed.py:74-79
    y = F.conv2d(x, weight, bias, stride=patch)
    return y.flatten(2).transpose(1, 2)


def interpolated_pos_embed(pos_embed, n_tokens, img_rows, img_cols, patch):
    """DINOv2.py:199-230.  ``pos_embed`` is [1, 1+M*M, D]; returns [1, n_tokens, D].
    Bicubic resize of the M x M grid with scale_factor=((h0+0.1)/M, (w0+0.1)/M)."""
    npatch = n_tokens - 1
    N = pos_embed.shape[1] - 1
    if npatch == N and img_rows == img_cols:
        return pos_embed
    pe = pos_embed.float()
    cls_pe, grid_pe = pe[:, 0], pe[:, 1:]
    dim = pe.shape[-1]
    m = int(math.sqrt(N))
    r0 = img_rows // patch + 0.1  # the reference names this w0 (DINOv2.py:209-213)
    c0 = img_cols // patch + 0.1
    grid = F.interpolate(
        grid_pe.reshape(1, m, m, dim).permute(0, 3, 1, 2),
        scale_factor=(r0 / math.sqrt(N), c0 / math.sqrt(N)),
        mode="bicubic",
    )
    assert int(r0) == grid.shape[-2] and int(c0) == grid.shape[-1]
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((cls_pe.unsqueeze(0), grid), dim=1)


def attention(x, w, prefix, heads):
    """dinov2/layers/attention.py:54-71 (the non-xFormers path MemEffAttention falls
    back to, :75-79)."""
    B, N, C = x.shape
    hd = C // heads
    qkv = F.linear(x, w[prefix + "qkv.weight"], w.get(prefix + "qkv.bias"))
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    p = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    y = (p @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(y, w[prefix + "proj.weight"], w.get(prefix + "proj.bias"))


def mlp(x, w, prefix):
    """dinov2/layers/mlp.py:34-40 (exact-erf GELU, nn.GELU default)."""
    y = F.gelu(F.linear(x, w[prefix + "fc1.weight"], w.get(prefix + "fc1.bias")))
    return F.linear(y, w[prefix + "fc2.weight"], w.get(prefix + "fc2.bias"))


def block(x, w, prefix, heads):
    """dinov2/layers/block.py:92-117, eval branch: x + ls1(attn(norm1 x)); x + ls2(mlp(norm2 x))."""
    C = x.shape[-1]
    a = attention(
        F.layer_norm(x, (C,), w[prefix + "norm1.weight"], w[prefix + "norm1.bias"], LN_EPS),
        w, prefix + "attn.", heads)
    if prefix + "ls1.gamma" in w:  # layer_scale.py:25-26
        a = a * w[prefix + "ls1.gamma"]
    x = x + a
    m = mlp(
        F.layer_norm(x, (C,), w[prefix + "norm2.weight"], w[prefix + "norm2.bias"], LN_EPS),
        w, prefix + "mlp.")
    if prefix + "ls2.gamma" in w:
        m = m * w[prefix + "ls2.gamma"]
    return x + m


def dinov2_features(image, w, *, patch, depth, heads, click_tokens=None,
                    injection="no_injection", prefix="", return_tokens=False):
    """DINOv2Featurizer.forward (DINOv2.py:500-546).

    image: [B,3,H,W] normalised; click_tokens: [B,h*w,D] or None.
    Returns [B,D,h,w] (a permuted view in the reference, :545)."""
    w = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)} if prefix else w
    B, _, H, W = image.shape
    h, wd = H // patch, W // patch
    D = w["cls_token"].shape[-1]
    inject = click_tokens is not None and injection != "no_injection"
    if inject and injection not in ("before_backbone", "after_backbone"):
        raise NameError(f"Unknown feats_injection_mode: {injection}")  # DINOv2.py:535-538

    x = patch_tokens(image, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], patch)
    if inject and injection == "before_backbone":  # DINOv2.py:518-523
        assert x.shape == click_tokens.shape
        x = x + click_tokens
    x = torch.cat((w["cls_token"].expand(B, -1, -1), x), dim=1)  # :525 / :240
    x = x + interpolated_pos_embed(w["pos_embed"], x.shape[1], H, W, patch)  # :526-528
    for i in range(depth):
        x = block(x, w, f"blocks.{i}.", heads)
    x = F.layer_norm(x, (D,), w["norm.weight"], w["norm.bias"], LN_EPS)  # :533
    feats = x[:, 1:]
    if inject and injection == "after_backbone":  # :509-516
        assert feats.shape == click_tokens.shape
        feats = feats + click_tokens
    if return_tokens:
        return feats
    return feats.reshape(-1, h, wd, D).permute(0, 3, 1, 2)  # :545


def dino_features(image, w, *, patch, depth, heads, feat_type="key", click_tokens=None,
                  injection="before_backbone", prefix=""):
    """DINOFeaturizer.forward (reference core/model/featurizers/DINO.py:529-611), n=1: a DINO-v1
    style ViT (no LayerScale, qkv bias, LN eps 1e-6) that returns either the last block's KEYS
    (``feat_type="key"``: channel index = d*heads + head, :588-603) or its normalised tokens."""
    w = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)} if prefix else w
    B, _, H, W = image.shape
    assert H % patch == 0 and W % patch == 0  # DINO.py:536-537
    h, wd = H // patch, W // patch
    D = w["cls_token"].shape[-1]
    x = patch_tokens(image, w["patch_embed.proj.weight"], w["patch_embed.proj.bias"], patch)
    if click_tokens is not None and injection == "before_backbone":  # :542-549
        assert x.shape == click_tokens.shape
        x = x + click_tokens
    x = torch.cat((w["cls_token"].expand(B, -1, -1), x), dim=1)
    x = x + interpolated_pos_embed(w["pos_embed"], x.shape[1], H, W, patch)  # DINO.py:289-314, same recipe
    keys = None
    for i in range(depth):
        p = f"blocks.{i}."
        if i == depth - 1:  # qkv of the last block (DINO.py:107-143 returns it)
            xn = F.layer_norm(x, (D,), w[p + "norm1.weight"], w[p + "norm1.bias"], LN_EPS)
            qkv = F.linear(xn, w[p + "attn.qkv.weight"], w.get(p + "attn.qkv.bias"))
            keys = qkv.reshape(B, -1, 3, heads, D // heads)[:, :, 1]  # [B, N, heads, hd]
        x = block(x, w, p, heads)
    feat = F.layer_norm(x, (D,), w["norm.weight"], w["norm.bias"], LN_EPS)
    if feat_type == "token":
        out = feat[:, 1:]
    elif feat_type == "key":
        out = keys[:, 1:].permute(0, 1, 3, 2).flatten(2)  # [B, T, hd*heads]: index d*heads + head
    else:
        raise ValueError("Unknown feat type:{}".format(feat_type))
    if click_tokens is not None and injection == "after_backbone":  # :572-580 / :590-597
        assert out.shape == click_tokens.shape
        out = out + click_tokens
    return out.reshape(B, h, wd, -1).permute(0, 3, 1, 2)


def simple_vit_tokens(x, w, *, patch, heads, dim_head=64, depth=None, prefix=""):
    """SimpleViTFeaturizer.forward (reference core/model/featurizers/simple_ViT.py:96-150): patches in
    (p1 p2 c) order -> LN -> Linear -> LN -> + 2-D sincos pos-emb -> pre-norm transformer -> LN."""
    w = {k[len(prefix):]: v for k, v in w.items() if k.startswith(prefix)} if prefix else w
    B, C, H, W = x.shape
    h, wd = H // patch, W // patch
    t = x.reshape(B, C, h, patch, wd, patch).permute(0, 2, 4, 3, 5, 1).reshape(B, h * wd, patch * patch * C)
    t = F.layer_norm(t, (t.shape[-1],), w["to_patch_embedding.1.weight"], w["to_patch_embedding.1.bias"])
    t = F.linear(t, w["to_patch_embedding.2.weight"], w["to_patch_embedding.2.bias"])
    dim = t.shape[-1]
    t = F.layer_norm(t, (dim,), w["to_patch_embedding.3.weight"], w["to_patch_embedding.3.bias"])
    # posemb_sincos_2d (simple_ViT.py:18-27)
    yy, xx = torch.meshgrid(torch.arange(h), torch.arange(wd), indexing="ij")
    omega = 1.0 / (10000 ** (torch.arange(dim // 4) / (dim // 4 - 1)))
    yy, xx = yy.flatten()[:, None] * omega[None, :], xx.flatten()[:, None] * omega[None, :]
    t = t + torch.cat((xx.sin(), xx.cos(), yy.sin(), yy.cos()), dim=1).float()
    i = 0
    while f"transformer.layers.{i}.0.to_qkv.weight" in w:
        p = f"transformer.layers.{i}."
        a = F.layer_norm(t, (dim,), w[p + "0.norm.weight"], w[p + "0.norm.bias"])
        q, k, v = F.linear(a, w[p + "0.to_qkv.weight"]).chunk(3, dim=-1)
        sp = lambda z: z.reshape(B, -1, heads, dim_head).transpose(1, 2)
        att = ((sp(q) @ sp(k).transpose(-1, -2)) * dim_head ** -0.5).softmax(-1) @ sp(v)
        t = F.linear(att.transpose(1, 2).reshape(B, -1, heads * dim_head), w[p + "0.to_out.weight"]) + t
        f = F.layer_norm(t, (dim,), w[p + "1.net.0.weight"], w[p + "1.net.0.bias"])
        f = F.linear(F.gelu(F.linear(f, w[p + "1.net.1.weight"], w[p + "1.net.1.bias"])), w[p + "1.net.3.weight"], w[p + "1.net.3.bias"])
        t = f + t
        i += 1
    return F.layer_norm(t, (dim,), w["transformer.norm.weight"], w["transformer.norm.bias"])


def maskclip_features(image, w, *, patch, heads, click_tokens=None, injection="no_injection", prefix="visual."):
    """MaskCLIPFeaturizer.forward (reference core/model/featurizers/MaskCLIP.py:41-92) on CLIP's
    VisionTransformer with patch_output=True (maskclip/model.py:321-358, :251-263): all but the last
    block run normally, the last block contributes only out_proj(v_proj(ln_1 x)); cls dropped; ln_post;
    @ proj.  fp32 here (the reference runs fp16 weights on the GPU)."""
    w = {k[len(prefix):]: v.float() for k, v in w.items() if k.startswith(prefix)}
    B, _, H, W = image.shape
    h, wd = H // patch, W // patch
    x = F.conv2d(image, w["conv1.weight"], None, stride=patch).flatten(2).permute(0, 2, 1)
    if click_tokens is not None and injection == "before_backbone":  # MaskCLIP.py:51-65
        assert x.shape == click_tokens.shape
        x = x + click_tokens
    D = x.shape[-1]
    x = torch.cat([w["class_embedding"].expand(B, 1, D), x], dim=1)
    pe = w["positional_embedding"]  # maskclip/interpolate.py:5-59: same bicubic recipe as DINO
    # Reference quirk: the before_backbone route (forward_without_patch_embed, model.py:389,402-404) passes
    # (h, w) = (H, W) where the normal route passes (w, h) = (H, W) (model.py:322,341): for non-square
    # images its grid is interpolated as [W/p, H/p] and then read row-major.  Reproduced as is.
    swap = click_tokens is not None and injection == "before_backbone"
    x = x + interpolated_pos_embed(pe.unsqueeze(0), x.shape[1], W if swap else H, H if swap else W, patch)[0]
    x = F.layer_norm(x, (D,), w["ln_pre.weight"], w["ln_pre.bias"])
    n = 0
    while f"transformer.resblocks.{n}.ln_1.weight" in w:
        n += 1
    hd = D // heads
    for i in range(n):
        p = f"transformer.resblocks.{i}."
        a = F.layer_norm(x, (D,), w[p + "ln_1.weight"], w[p + "ln_1.bias"])
        if i == n - 1:  # forward_v (model.py:251-263)
            v = F.linear(a, w[p + "attn.in_proj_weight"][-D:], w[p + "attn.in_proj_bias"][-D:])
            x = F.linear(v, w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
            break
        qkv = F.linear(a, w[p + "attn.in_proj_weight"], w[p + "attn.in_proj_bias"]).reshape(B, -1, 3, heads, hd)
        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :, 1].transpose(1, 2), qkv[:, :, 2].transpose(1, 2)
        o = ((q * hd ** -0.5) @ k.transpose(-1, -2)).softmax(-1) @ v
        x = x + F.linear(o.transpose(1, 2).reshape(B, -1, D), w[p + "attn.out_proj.weight"], w[p + "attn.out_proj.bias"])
        m = F.linear(F.layer_norm(x, (D,), w[p + "ln_2.weight"], w[p + "ln_2.bias"]), w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"])
        x = x + F.linear(m * torch.sigmoid(1.702 * m), w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
    x = F.layer_norm(x[:, 1:], (D,), w["ln_post.weight"], w["ln_post.bias"]) @ w["proj"]
    if click_tokens is not None and injection == "after_backbone":  # MaskCLIP.py:75-83
        assert x.shape == click_tokens.shape
        x = x + click_tokens
    return x.reshape(B, h, wd, -1).permute(0, 3, 1, 2)
