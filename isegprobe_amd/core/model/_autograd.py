"""torch.autograd glue for the trainable tail of the path (seg head, resize, click patch-embed,
after-backbone click injection).  Forward and backward are HIP kernels; autograd only carries the
graph so that the reference's training loop (``loss.backward()``, Adam; core/training/trainer.py:
219-226) runs unchanged.  Gradients of parameters are fp32; activation gradients bf16 NHWC.

Scope: gradients flow logits -> head -> resize -> features -> (frozen ViT blocks, when the click
tokens are injected ``before_backbone`` -- the reference's default training mode,
models/sbd/dinov2/patch-embed_*.py:40) -> click tokens -> embed_coords.  Frozen weights get no
gradient; only activation gradients are propagated through the trunk (``ViTTrunkFn``).  The learned
upsamplers (LoftUp / LiFT / FeatUp JBU) have no backward yet: with them the model raises when asked
to train with ``before_backbone``."""
import torch

from ... import hip_ops as ops

BF16 = torch.bfloat16


def _pack_conv(weight):
    n = weight.shape[0]
    return weight.detach().permute(0, 2, 3, 1).reshape(n, -1).to(BF16).contiguous()


# ---- test hook: ReLU masks imposed from outside (tests/test_training_gpu.py).  A 16-bit forward puts a pre-activation that is
# within rounding noise of zero on either side of it, and one flipped mask entry moves a weight-gradient sum by its whole
# term; comparing gradients against the fp32 oracle WITH THE ORACLE'S MASKS separates that (legitimate) effect from errors of
# the backward kernels.  The list holds {0,1}-valued [B,H,W,N] tensors in backward order; each conv + ReLU node of the head
# takes the next one instead of (its own output > 0).  None outside the tests.
_RELU_MASKS = None


class impose_relu_masks:
    def __init__(self, masks_in_backward_order):
        self.masks = list(masks_in_backward_order)

    def __enter__(self):
        global _RELU_MASKS
        _RELU_MASKS = self.masks
        return self

    def __exit__(self, *exc):
        global _RELU_MASKS
        _RELU_MASKS = None


def _imposed_mask(y):
    if not _RELU_MASKS:
        return None
    m = _RELU_MASKS.pop(0).to(device=y.device, dtype=y.dtype)
    if m.shape[:-1] == y.shape[:-1] and m.shape[-1] < y.shape[-1]:  # channel-padded maps (LiFT): the padding stays masked
        m = torch.nn.functional.pad(m, (0, y.shape[-1] - m.shape[-1]))
    if m.shape != y.shape:
        raise RuntimeError(f"imposed ReLU mask {tuple(m.shape)} does not fit the node's output {tuple(y.shape)}")
    return m.contiguous()


class Conv3x3ReluFn(torch.autograd.Function):
    """relu(conv3x3(x) + bias): x [B,H,W,C] bf16 NHWC, weight [N,C,3,3] fp32 -> [B,H,W,N] bf16."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        y = ops.conv3x3(x, _pack_conv(weight), bias.detach().float().contiguous(), "relu")
        ctx.save_for_backward(x, y, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight = ctx.saved_tensors
        m = _imposed_mask(y)
        g, db = ops.relu_mask_colsum(gy.contiguous(), y if m is None else m)
        dx, dweight = _conv3x3_grads(x, g, weight, ctx.needs_input_grad[0])
        return dx, dweight, db


def _conv3x3_grads(x, g, weight, need_dx):
    """Weight (and optionally data) gradient of a 3x3 conv given the pre-activation gradient g."""
    B, H, W, C = x.shape
    N = weight.shape[0]
    M = B * H * W
    dw = ops.conv3x3_wgrad(g, x)  # all nine taps from one staged patch per 8x8-pixel tile
    dweight = dw.view(N, 3, 3, C).permute(0, 3, 1, 2).contiguous()
    dx = None
    if need_dx:  # data gradient = the same conv kernel with rotated, transposed weights
        w_rot = weight.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * N).to(BF16).contiguous()
        dx = ops.conv3x3(g, w_rot, None, None)
    return dx, dweight


class Conv3x3OfBilinearReluFn(torch.autograd.Function):
    """relu(conv3x3(bilinear_ac(x, (H, W))) + bias) without the resized map, forward AND backward (iseg_probe_model.py:120-129 +
    conv_heads.py:59-73 under autograd): x [B,h,w,C] bf16 NHWC low-resolution features, weight [N,C,3,3] fp32 -> [B,H,W,N] bf16.
    Forward: Z = x [W_0 .. W_8]^T (one GEMM on IEEE-half operands) and the blend.  Backward: the ReLU-masked gradient goes
    through the blend's adjoint to dZ [B h w, 9 N]; dx = dZ Wz and dWz = dZ^T x are low-resolution GEMMs -- no full-resolution
    data-gradient conv, no full-resolution weight-gradient reduction."""

    @staticmethod
    def forward(ctx, x, weight, bias, H, W):
        B, h, w, C = x.shape
        N = weight.shape[0]
        wz32 = weight.detach().float().permute(2, 3, 0, 1).reshape(9 * N, C)  # rows t*N + n, t = ky*3 + kx
        z = ops.linear(ops.to_f16(x).view(B * h * w, C), wz32.to(ops.F16).contiguous())
        y = ops.conv3x3_of_bilinear_blend(z, bias.detach().float().contiguous(), B, h, w, H, W, N, relu=True, out_dtype=BF16)
        ctx.save_for_backward(x, y, weight)
        ctx.geom = (B, h, w, H, W, C, N)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight = ctx.saved_tensors
        B, h, w, H, W, C, N = ctx.geom
        m = _imposed_mask(y)
        g, db = ops.relu_mask_colsum(gy.contiguous(), y if m is None else m)
        dz = ops.conv3x3_of_bilinear_blend_bwd(g, B, h, w, H, W, N)
        wz = weight.detach().float().permute(2, 3, 0, 1).reshape(9 * N, C)
        dx = None
        if ctx.needs_input_grad[0]:  # dx = dZ Wz: [B h w, 9N] x [9N, C]
            dx = ops.linear(dz, wz.t().contiguous().to(BF16)).view(B, h, w, C)
        dwz, _ = ops.linear_wgrad(dz, x.reshape(B * h * w, C), want_bias=False)  # [9N, C] f32
        dweight = dwz.view(3, 3, N, C).permute(2, 3, 0, 1).contiguous()
        return dx, dweight, db, None, None


class Conv3x3ReluClassifierFn(torch.autograd.Function):
    """Last 3x3 conv + ReLU + 1x1 classifier of ConvSegHead (conv_heads.py:69-73) as one node: the
    classifier backward produces the conv's pre-activation gradient already ReLU-masked together with
    its column sums (the conv's bias gradient), so no separate mask pass runs."""

    @staticmethod
    def forward(ctx, x, weight, bias, cls_weight, cls_bias):
        y = ops.conv3x3(x, _pack_conv(weight), bias.detach().float().contiguous(), "relu")
        B, H, W, _ = y.shape
        wc = cls_weight.detach().float().reshape(-1).contiguous()
        out = ops.classifier(y, wc, float(cls_bias.detach().float().item())).view(B, 1, H, W)
        ctx.save_for_backward(x, y, weight, cls_weight)
        return out

    @staticmethod
    def backward(ctx, gl):
        x, y, weight, cls_weight = ctx.saved_tensors
        wc = cls_weight.detach().float().reshape(-1).contiguous()
        g, dwc, dbc, db = ops.classifier_bwd(gl.float().reshape(-1), y, wc, want_dx_colsum=True)
        m = _imposed_mask(y)
        if m is not None:  # (test hook) the pre-activation gradient gl (x) wc under the imposed mask instead of (y > 0)
            gfull = (gl.float().reshape(-1, 1) * wc.reshape(1, -1)).to(y.dtype).view_as(y)
            g, db = ops.relu_mask_colsum(gfull.contiguous(), m)
        dx, dweight = _conv3x3_grads(x, g, weight, ctx.needs_input_grad[0])
        return dx, dweight, db, dwc.view_as(cls_weight), dbc.view(1)


class Conv1x1ReluFn(torch.autograd.Function):
    """relu(x W^T + b) per pixel (SimpleConvSegHead layers)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, H, W, C = x.shape
        w = weight.detach().flatten(1).to(BF16).contiguous()
        y = ops.linear(x.view(-1, C), w, bias.detach().float().contiguous(), "relu").view(B, H, W, -1)
        ctx.save_for_backward(x, y, weight)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, y, weight = ctx.saved_tensors
        B, H, W, C = x.shape
        N = weight.shape[0]
        M = B * H * W
        g, db = ops.relu_mask_colsum(gy.contiguous(), y)
        dw = torch.zeros(N, C, device=x.device, dtype=torch.float32)
        ops.tn_gemm_atomic(g.view(M, N), x.view(M, C), dw)
        dx = None
        if ctx.needs_input_grad[0]:
            wt = weight.detach().flatten(1).t().to(BF16).contiguous()  # [C, N]
            dx = ops.linear(g.view(M, N), wt, None, None).view(B, H, W, C)
        return dx, dw.view_as(weight), db


class ClassifierFn(torch.autograd.Function):
    """1x1 conv C -> 1 on an NHWC bf16 map -> logits [B,1,H,W] fp32.  ``post_relu``: x is the output of a conv+ReLU
    layer, whose ReLU mask the backward kernel then applies to dx in the same pass; for a signed x (the "linear"
    head, a conv head with num_layers = 0) dx is the plain g * w."""

    @staticmethod
    def forward(ctx, x, weight, bias, post_relu=False):
        ctx.post_relu = bool(post_relu)
        B, H, W, C = x.shape
        w = weight.detach().float().reshape(-1).contiguous()
        out = ops.classifier(x, w, float(bias.detach().float().item())).view(B, 1, H, W)
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, gl):
        x, weight = ctx.saved_tensors
        w = weight.detach().float().reshape(-1).contiguous()
        dx, dw, db = ops.classifier_bwd(gl.float().reshape(-1), x, w, relu_mask=ctx.post_relu)
        # NB: with post_relu dx carries the ReLU mask of x; the producing layer's backward masks again with the
        # same mask, which is idempotent.
        return (dx if ctx.needs_input_grad[0] else None), dw.view_as(weight), db.view(1), None


class ResizeBilinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, H, W):
        ctx.hw = x.shape[1:3]
        return ops.resize_nhwc(x, H, W, "bilinear")

    @staticmethod
    def backward(ctx, gy):
        return ops.resize_bilinear_nhwc_bwd(gy.contiguous(), ctx.hw[0], ctx.hw[1]), None, None


class ResizeLogitsFn(torch.autograd.Function):
    """Final logits resize [B,1,h,w] -> [B,1,H,W] fp32 (iseg_base_model.py:75-80)."""

    @staticmethod
    def forward(ctx, x, H, W):
        ctx.hw = x.shape[2:]
        return ops.resize_bilinear_nchw_f32(x.float().contiguous(), H, W)

    @staticmethod
    def backward(ctx, gy):
        return ops.resize_bilinear_nchw_f32_bwd(gy.float().contiguous(), ctx.hw[0], ctx.hw[1]), None, None


class TokenAddFn(torch.autograd.Function):
    """features [B,T,D] bf16 (frozen backbone output) + click tokens [B,T,D] fp32 (DINOv2.py:516)."""

    @staticmethod
    def forward(ctx, feats, clicks):
        out = feats.clone()
        B, T, D = clicks.shape
        ops.token_add_(out.view(-1, D), clicks.detach().float().contiguous(), B, T, has_cls=False)
        return out

    @staticmethod
    def backward(ctx, gy):
        return None, gy.float()


class PatchEmbedFn(torch.autograd.Function):
    """tokens = patchify(maps) . W^T + b (featurizers/utils/patch_embed.py:37-42); the maps (click
    disks, previous mask) carry no gradient."""

    @staticmethod
    def forward(ctx, maps, weight, bias, patch, kpad):
        D = weight.shape[0]
        K = weight[0].numel()
        wp = torch.zeros(D, kpad, device=weight.device, dtype=BF16)
        wp[:, :K] = weight.detach().flatten(1).to(BF16)
        A = ops.patchify(maps.float().contiguous(), None, None, patch, kpad)
        tok = ops.linear(A, wp, bias.detach().float().contiguous(), None, out_dtype=torch.float32)
        ctx.save_for_backward(A, weight)
        B, _, H, W = maps.shape
        return tok.view(B, (H // patch) * (W // patch), D)

    @staticmethod
    def backward(ctx, gtok):
        A, weight = ctx.saved_tensors
        D = weight.shape[0]
        K = weight[0].numel()
        g = gtok.reshape(-1, D).to(BF16).contiguous()
        dw = torch.zeros(D, A.shape[1], device=A.device, dtype=torch.float32)
        ops.tn_gemm_atomic(g, A, dw)
        ones = torch.ones(g.shape[0], 8, device=g.device, dtype=BF16)
        db = torch.zeros(D, 8, device=g.device, dtype=torch.float32)
        ops.tn_gemm_atomic(g, ones, db)
        return None, dw[:, :K].reshape(weight.shape).contiguous(), db[:, 0].contiguous(), None, None


def _block_forward_saving(x, blk, heads, B, L, eps, act):
    """One pre-LN transformer block on the fp32 residual stream x (updated in place), keeping what its backward
    needs.  blk keys: n1w n1b qkv_w qkv_b proj_w proj_b ls1 n2w n2b fc1_w fc1_b fc2_w fc2_b ls2."""
    x_in = x.clone()
    h1 = ops.layernorm(x, blk["n1w"], blk["n1b"], eps)
    qkv = ops.linear(h1, blk["qkv_w"], blk["qkv_b"])
    att, lse = ops.attention_packed_qkv_lse(qkv, B, L, heads, 64 ** -0.5)
    ops.linear_residual_(x, att, blk["proj_w"], blk["proj_b"], blk["ls1"])
    x_mid = x.clone()
    h2 = ops.layernorm(x, blk["n2w"], blk["n2b"], eps)
    hid, pre = ops.linear_gelu_save(h2, blk["fc1_w"], blk["fc1_b"], act)
    ops.linear_residual_(x, hid, blk["fc2_w"], blk["fc2_b"], blk["ls2"])
    return (x_in, qkv, att, lse, x_mid, pre)


def _block_bwd_weights(blk):
    """Transposed (data-gradient) weight copies, LayerScale folded in, cached beside the packed weights."""
    if "bwd" not in blk:
        def t(w, gamma=None):  # forward y = x W^T (W [N,K]); backward gx = gy W -> gemm weight [K,N]
            w = w.float()
            if gamma is not None:
                w = w * gamma[:, None]
            return w.t().contiguous().to(BF16)
        blk["bwd"] = dict(fc2=t(blk["fc2_w"], blk["ls2"]), fc1=t(blk["fc1_w"]), proj=t(blk["proj_w"], blk["ls1"]),
                          qkv=t(blk["qkv_w"]))
    return blk["bwd"]


def _block_backward(gx, g16, blk, saved, heads, B, L, eps, act):
    """Activation gradient through one block: (gx fp32 stream gradient, its bf16 copy) at the block output ->
    the same pair at the block input."""
    x_in, qkv, att, lse, x_mid, pre = saved
    W = _block_bwd_weights(blk)
    g_pre = ops.linear_mul_dgelu(g16, W["fc2"], pre, act)             # d/d(fc1 out), LayerScale + act' fused
    g_h2 = ops.linear(g_pre, W["fc1"])
    gx, g16 = ops.layernorm_bwd(x_mid, g_h2, blk["n2w"], eps, gx=gx)
    g_att = ops.linear(g16, W["proj"])
    g_qkv = ops.attention_packed_qkv_bwd(qkv, att, g_att, lse, B, L, heads, 64 ** -0.5)
    g_h1 = ops.linear(g_qkv, W["qkv"])
    return ops.layernorm_bwd(x_in, g_h1, blk["n1w"], eps, gx=gx)


class ViTTrunkFn(torch.autograd.Function):
    """Frozen DINOv2-style trunk on the residual stream: ``x0`` [B*(T+1), D] fp32 (patch + cls +
    pos-embed tokens, click tokens already added on rows 1..T of each image) -> final-norm features
    [B*T, D] bf16 (cls dropped).  Forward = the inference kernels plus the statistics the backward
    needs (attention log-sum-exp, fc1 pre-activations, the residual stream before each norm);
    backward = activation gradients only (block.py:92-117, attention.py:54-71, mlp.py:34-40 under
    autograd with all weights frozen):
        LN bwd -> fc2^T (x LayerScale, x gelu') -> fc1^T -> LN bwd -> proj^T (x LayerScale) ->
        attention bwd -> qkv^T -> LN bwd, accumulating into an fp32 gradient stream."""

    @staticmethod
    def forward(ctx, x0, packed, heads, B, T, eps, last_keys=False):
        """last_keys: stop inside the last block and return its attention KEYS (cls dropped, channel = d*heads + head),
        the dense feature the reference's DINO ViT featurizer uses (DINO.py:582-590, feat_type="key")."""
        L = T + 1
        x = x0.detach().clone()
        saved = []
        nblk = len(packed["blocks"])
        for i, blk in enumerate(packed["blocks"]):
            if last_keys and i == nblk - 1:
                D = heads * 64
                h1 = ops.layernorm(x, blk["n1w"], blk["n1b"], eps)
                qkv = ops.linear(h1, blk["qkv_w"], blk["qkv_b"])
                k = qkv.view(B, L, 3, heads, 64)[:, 1:, 1]
                ctx.saved, ctx.x_final, ctx.packed, ctx.geom = saved, x.clone(), packed, (heads, B, T, eps, True)
                return k.permute(0, 1, 3, 2).reshape(B * T, D).contiguous()
            saved.append(_block_forward_saving(x, blk, heads, B, L, eps, "gelu"))
        feats = ops.layernorm(x, packed["nw"], packed["nb"], eps, group_out=T, skip=1, rows_out=B * T)
        ctx.saved, ctx.x_final, ctx.packed, ctx.geom = saved, x, packed, (heads, B, T, eps, False)
        return feats

    @staticmethod
    def backward(ctx, gfeats):
        heads, B, T, eps, last_keys = ctx.geom
        L = T + 1
        P = ctx.packed
        blocks = P["blocks"]
        if last_keys:
            # keys = LN1(x) Wk^T + bk of the last block, channel-permuted with the cls row dropped
            D = heads * 64
            blk = blocks[-1]
            if "bwd_k" not in blk:
                blk["bwd_k"] = blk["qkv_w"].float()[D:2 * D].t().contiguous().to(BF16)
            g_k = torch.zeros(B, L, D, device=gfeats.device, dtype=BF16)
            g_k[:, 1:] = gfeats.contiguous().view(B, T, 64, heads).permute(0, 1, 3, 2).reshape(B, T, D)
            g_h1 = ops.linear(g_k.view(B * L, D), blk["bwd_k"])
            gx, g16 = ops.layernorm_bwd(ctx.x_final, g_h1, blk["n1w"], eps)
            blocks = blocks[:-1]
        else:
            gx, g16 = ops.layernorm_bwd(ctx.x_final, gfeats.contiguous().view(B * T, -1), P["nw"], eps, group_out=T, skip=1)
        for blk, sv in zip(reversed(blocks), reversed(ctx.saved)):
            gx, g16 = _block_backward(gx, g16, blk, sv, heads, B, L, eps, "gelu")
        ctx.saved = None
        return gx, None, None, None, None, None, None


class MaskCLIPTrunkFn(torch.autograd.Function):
    """Frozen CLIP ViT trunk of the MaskCLIP featurizer (maskclip/model.py:251-263,321-430): tokens (+ clicks, cls,
    pos-embed) -> ln_pre -> L-1 residual blocks (QuickGELU) -> value path of the last block (out_proj(v_proj(ln_1 x)),
    no residual) -> cls drop + ln_post -> projection.  Activation gradients only (all weights frozen)."""

    @staticmethod
    def forward(ctx, x0, P, heads, B, T, out_dim):
        L, eps = T + 1, 1e-5
        x0 = x0.detach()
        x = ops.layernorm(x0, P["pre_w"], P["pre_b"], eps, out_dtype=torch.float32)
        saved = [_block_forward_saving(x, blk, heads, B, L, eps, "quick_gelu") for blk in P["blocks"][:-1]]
        last = P["blocks"][-1]
        a = ops.layernorm(x, last["n1w"], last["n1b"], eps)
        vout = ops.linear(ops.linear(a, last["v_w"], last["v_b"]), last["proj_w"], last["proj_b"])
        post = ops.layernorm(vout, P["post_w"], P["post_b"], eps, group_out=T, skip=1, rows_out=B * T)
        feats = ops.linear(post, P["proj"], None)
        ctx.saved, ctx.tensors, ctx.P, ctx.geom = saved, (x0, x, vout), P, (heads, B, T, out_dim)
        return feats[:, :out_dim].contiguous()

    @staticmethod
    def backward(ctx, gfeats):
        heads, B, T, out_dim = ctx.geom
        L, eps, P = T + 1, 1e-5, ctx.P
        x0, x_last, vout = ctx.tensors
        last = P["blocks"][-1]
        if "bwd_tail" not in P:
            t = lambda w: w.float().t().contiguous().to(BF16)
            P["bwd_tail"] = dict(proj=t(P["proj"]), out=t(last["proj_w"]), v=t(last["v_w"]))
        Wt = P["bwd_tail"]
        n_out = P["proj"].shape[0]
        g = torch.zeros(B * T, n_out, device=gfeats.device, dtype=BF16)
        g[:, :out_dim] = gfeats
        g_post = ops.linear(g, Wt["proj"])
        _, g_vout = ops.layernorm_bwd(vout, g_post, P["post_w"], eps, group_out=T, skip=1)
        g_a = ops.linear(ops.linear(g_vout, Wt["out"]), Wt["v"])
        gx, g16 = ops.layernorm_bwd(x_last, g_a, last["n1w"], eps)
        for blk, sv in zip(reversed(P["blocks"][:-1]), reversed(ctx.saved)):
            gx, g16 = _block_backward(gx, g16, blk, sv, heads, B, L, eps, "quick_gelu")
        g0, _ = ops.layernorm_bwd(x0, g16, P["pre_w"], eps, want_bf16=False)  # ln_pre
        ctx.saved = ctx.tensors = None
        return g0, None, None, None, None, None


class TokenInjectFn(torch.autograd.Function):
    """x0[b, 1+t] = tokens[b, 1+t] + clicks[b, t] (DINOv2.py:518-523): the gradient of the click tokens is the
    gradient of the stream rows 1..T."""

    @staticmethod
    def forward(ctx, xs, clicks, B, T):
        out = xs.clone()
        ops.token_add_(out, clicks.detach().float().contiguous(), B, T, has_cls=True)
        ctx.geom = (B, T)
        return out

    @staticmethod
    def backward(ctx, gx):
        B, T = ctx.geom
        return None, gx.view(B, T + 1, -1)[:, 1:].contiguous(), None, None


def grad_mode(*modules_or_params):
    """True when autograd should record: grad enabled and something trainable is involved."""
    if not torch.is_grad_enabled():
        return False
    for m in modules_or_params:
        ps = m.parameters() if hasattr(m, "parameters") else [m]
        if any(p.requires_grad for p in ps):
            return True
    return False
