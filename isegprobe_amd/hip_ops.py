"""Thin torch-tensor wrappers over the C-ABI (include/isegprobe_hip.h).

torch is used here only for device memory and the current stream; every arithmetic op on
the path is a HIP kernel in libisegprobe_hip.so.  All functions raise ``IspError`` on a
non-zero return code and require CUDA(HIP) tensors -- there is no CPU fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import Epilogue, IspError, check

BF16 = torch.bfloat16
F16 = torch.float16  # IEEE half: the FeatUp-JBU stack, the seg head's inference convolutions, LoftUp's inference stream


def _stream():
    # the caller's current stream (graph capture included); the raw accessor is ~8x cheaper than
    # torch.cuda.current_stream() and this runs once per launch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _need(t, dtype, name, contiguous=True):
    if not t.is_cuda:
        raise IspError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise IspError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise IspError(f"{name}: expected a contiguous tensor")
    return t


def click_maps(points, H, W, norm_radius, spatial_scale=1.0, use_disks=False, round_clicks=False, out=None):
    """points [B,2P,3] f32 -> [B,2,H,W] f32 (reference core/model/ops.py:35-77)."""
    points = _need(points.float().contiguous(), torch.float32, "points")
    B, P2, three = points.shape
    if three != 3 or P2 % 2:
        raise IspError("points must be [B, 2P, 3]")
    if out is None:
        out = torch.empty(B, 2, H, W, device=points.device, dtype=torch.float32)
    check(_lib.lib().isp_click_maps_fwd(_p(points), _p(out), B, P2 // 2, H, W, float(norm_radius),
                                        float(spatial_scale), int(use_disks), int(round_clicks), _stream()),
          "isp_click_maps_fwd")
    return out


def normalize(image, mean, std, want_prev=True):
    """image [B,3|4,H,W] f32 -> (normalised [B,3,H,W], prev_mask [B,1,H,W] or None)."""
    image = _need(image, torch.float32, "image")
    B, C, H, W = image.shape
    out = torch.empty(B, 3, H, W, device=image.device, dtype=torch.float32)
    prev = torch.empty(B, 1, H, W, device=image.device, dtype=torch.float32) if (C == 4 and want_prev) else None
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().isp_normalize_fwd(_p(image), _p(out), _p(prev), B, C, H, W, m, s, _stream()), "isp_normalize_fwd")
    return out, prev


def patchify(image, prev, maps, patch, Kpad):
    """Rows of [image | prev | maps] patches (channel-major, then i, j), zero-padded to Kpad, bf16.
    Any of the three NCHW f32 inputs may be None."""
    srcs = [t for t in (image, prev, maps) if t is not None]
    if not srcs:
        raise IspError("patchify needs at least one input")
    for t in srcs:
        _need(t, torch.float32, "patchify input")
    B, _, H, W = srcs[0].shape
    nch = [0 if t is None else t.shape[1] for t in (image, prev, maps)]
    A = torch.empty(B * (H // patch) * (W // patch), Kpad, device=srcs[0].device, dtype=BF16)
    check(_lib.lib().isp_patchify_fwd(_p(image), _p(prev), _p(maps), _p(A), B, H, W, patch, nch[0], nch[1], nch[2],
                                      Kpad, _stream()), "isp_patchify_fwd")
    return A


def _epilogue(kind, out, ldo=0, bias=None, gamma=None, pos=None, tokens=0, res=None, alpha=0.0):
    ep = Epilogue()
    ep.kind = kind
    ep.out = out.data_ptr()
    ep.ldo = ldo
    ep.bias = bias.data_ptr() if bias is not None else None
    ep.gamma = gamma.data_ptr() if gamma is not None else None
    ep.pos = pos.data_ptr() if pos is not None else None
    ep.tokens_per_image = tokens
    ep.res = res.data_ptr() if res is not None else None
    ep.alpha = alpha
    ep.out2 = None
    ep.out3 = None
    return ep


def gemm(A, Wt, ep, M=None):
    """A [M,K] bf16 (row stride A.stride(0)), Wt [N,K] bf16, ep from _epilogue.  Both IEEE half: the f16 engine (bias,
    bias + GELU and residual epilogues only; 16-bit outputs are half)."""
    half = A.dtype == F16
    _need(A, F16 if half else BF16, "A", contiguous=False)
    _need(Wt, F16 if half else BF16, "Wt")
    if A.stride(1) != 1:
        raise IspError("A must be row-major")
    N, K = Wt.shape
    M = A.shape[0] if M is None else M
    fn, name = (_lib.lib().isp_gemm_f16, "isp_gemm_f16") if half else (_lib.lib().isp_gemm_bf16, "isp_gemm_bf16")
    check(fn(_p(A), A.stride(0), _p(Wt), M, N, K, ctypes.byref(ep), _stream()), name)


def linear(A, Wt, bias=None, act=None, out_dtype=None):
    """bf16 (or f16) A [M,K] x Wt[N,K]^T + bias, optional 'relu'/'gelu'; returns [M,N] in A's dtype (or f32 for bf16 A)."""
    N = Wt.shape[0]
    out_dtype = A.dtype if out_dtype is None else out_dtype
    out = torch.empty(A.shape[0], N, device=A.device, dtype=out_dtype)
    if out_dtype == A.dtype:
        kind = {None: _lib.EP_BIAS_BF16, "relu": _lib.EP_BIAS_RELU_BF16, "gelu": _lib.EP_BIAS_GELU_BF16,
                "quick_gelu": _lib.EP_BIAS_QGELU_BF16}[act]
    else:
        if act is not None:
            raise IspError("fp32 output supports no activation")
        kind = _lib.EP_BIAS_F32
    gemm(A, Wt, _epilogue(kind, out, N, bias))
    return out


def linear_gelu_save(A, Wt, bias, act="gelu"):
    """Training forward of Mlp.fc1 (mlp.py:34-40): returns (act(pre), pre), pre = A Wt^T + bias, both bf16;
    act = "gelu" (exact erf) or "quick_gelu" (CLIP)."""
    N = Wt.shape[0]
    out = torch.empty(A.shape[0], N, device=A.device, dtype=BF16)
    pre = torch.empty_like(out)
    ep = _epilogue({"gelu": _lib.EP_BIAS_GELU_SAVE_BF16, "quick_gelu": _lib.EP_BIAS_QGELU_SAVE_BF16}[act], out, N, bias)
    ep.out2 = pre.data_ptr()
    gemm(A, Wt, ep)
    return out, pre


def linear_mul_dgelu(A, Wt, pre, act="gelu"):
    """(A Wt^T) * act'(pre): the fc2 data gradient with the GELU / QuickGELU backward fused; pre bf16 [M,N]."""
    _need(pre, BF16, "pre")
    N = Wt.shape[0]
    if tuple(pre.shape) != (A.shape[0], N):
        raise IspError("pre-activation shape mismatch")
    out = torch.empty(A.shape[0], N, device=A.device, dtype=BF16)
    gemm(A, Wt, _epilogue({"gelu": _lib.EP_MUL_DGELU_BF16, "quick_gelu": _lib.EP_MUL_DQGELU_BF16}[act], out, N, res=pre))
    return out


def linear_residual_(x, A, Wt, bias=None, gamma=None, ldo=None):
    """x (f32 [M,N]) += gamma * (A Wt^T + bias), in place.  ``ldo``: row stride of x in elements when the M rows are a
    strided subset of a larger matrix (the ViT's class-token rows: every (T+1)-th row of the stream)."""
    _need(x, torch.float32, "x")
    gemm(A, Wt, _epilogue(_lib.EP_RESIDUAL_F32, x, x.shape[-1] if ldo is None else ldo, bias, gamma))
    return x


def linear_residual_stats_(x, A, Wt, bias=None, gamma=None):
    """linear_residual_ on IEEE-half operands that also returns (x16, stats): the half copy of the updated fp32 rows and their
    per-row partial sums f32 [slots, M, 2] -- what ``linear_lnfold`` needs to run the block's next LayerNorm + Linear as one
    GEMM (csrc/gemm.hip, EpResidualStats / EpLnFold)."""
    _need(x, torch.float32, "x")
    _need(A, F16, "A")
    M, N = x.shape
    x16 = torch.empty(M, N, device=x.device, dtype=F16)
    stats = torch.empty(_lib.lib().isp_gemm_f16_stats_slots(M, N), M, 2, device=x.device, dtype=torch.float32)
    ep = _epilogue(_lib.EP_RESIDUAL_STATS_F32, x, N, bias, gamma)
    ep.out2, ep.out3 = stats.data_ptr(), x16.data_ptr()
    gemm(A, Wt, ep)
    return x16, stats


def gemm_f16_stats_slots(M, N):
    """Partial-statistics slots ``linear_residual_stats_`` writes for an M x N problem (follows the tile configuration)."""
    return int(_lib.lib().isp_gemm_f16_stats_slots(M, N))


def vit_mlp_pack(norm_w, norm_b, fc1_w, fc1_b, fc2_w, fc2_b, ls=None, dtype=BF16):
    """Weights of isp_vit_mlp_fused from a block's fp32 parameters: LayerNorm affine folded into fc1, LayerScale into
    fc2, fc2's hidden axis permuted inside every group of 16 to the 32x32 accumulator order of the first product
    (physical 8g+e <- logical 4g+e for e < 4, 8+4g+(e-4) otherwise, g in {0,1})."""
    w1 = fc1_w.float() * norm_w.float()[None, :]
    b1 = fc1_b.float() + fc1_w.float() @ norm_b.float()
    w2, b2 = fc2_w.float(), fc2_b.float()
    if ls is not None:
        w2, b2 = w2 * ls.float()[:, None], b2 * ls.float()
    hid = w2.shape[1]
    perm = torch.tensor([(4 * (p // 8) + p % 8) if p % 8 < 4 else (8 + 4 * (p // 8) + p % 8 - 4) for p in range(16)],
                        device=w2.device)
    idx = (torch.arange(0, hid, 16, device=w2.device)[:, None] + perm[None, :]).reshape(-1)
    return (w1.to(dtype).contiguous(), b1.contiguous(), w2[:, idx].to(dtype).contiguous(), b2.contiguous())


def vit_mlp_fused_(x, w1, b1, w2p, b2, eps):
    """In place on the fp32 residual stream x [M, D]:  x += ls * fc2(GELU(fc1(LayerNorm(x))))  (weights from vit_mlp_pack)."""
    _need(x, torch.float32, "x")
    M, D = x.shape
    check(_lib.lib().isp_vit_mlp_fused(_p(x), _p(w1), _p(b1), _p(w2p), _p(b2), M, D, w1.shape[0], float(eps), _stream()),
          "isp_vit_mlp_fused")
    return x


def vit_mlp_fused_rows_(x, w1, b1, w2p, b2, eps, images, rows_per_image, first_row, T):
    """``vit_mlp_fused_`` over rows first_row .. first_row + T - 1 of every image's rows_per_image rows (the patch tokens;
    the class-token rows are the caller's)."""
    _need(x, torch.float32, "x")
    M, D = x.shape
    if M != images * rows_per_image:
        raise IspError("vit_mlp_fused_rows_: x must hold images * rows_per_image rows")
    if w1.dtype != w2p.dtype or w1.dtype not in (BF16, F16):
        raise IspError("vit_mlp_fused_rows_: w1 / w2p must both be bf16 or both IEEE half")
    check(_lib.lib().isp_vit_mlp_fused_rows(_p(x), _p(w1), _p(b1), _p(w2p), _p(b2), images, rows_per_image, first_row, T, D,
                                            w1.shape[0], float(eps), _DTYPE_CODE[w1.dtype], _stream()), "isp_vit_mlp_fused_rows")
    return x


def vit_mlp_fused_supported(D, hid):
    return (D, hid) == (384, 1536)


def conv_takes_f16(N):
    """Output-channel counts the f16 patch conv handles: 192-channel blocks, or 128-channel blocks wasting <= 15 %."""
    return N % 192 == 0 or (N >= 128 and ((N + 127) // 128 * 128 - N) * 100 <= 15 * N)


def _conv_entry(x, Wt):
    """The conv entry point for an operand pair: both bf16, or both IEEE half (the head behind FeatUp JBU)."""
    if x.dtype == F16:
        _need(x, F16, "x")
        _need(Wt, F16, "Wt")
        if not conv_takes_f16(Wt.shape[0]):
            raise IspError("the f16 conv needs output channels that tile into 192- or (nearly) 128-channel blocks")
        return _lib.lib().isp_conv3x3_nhwc_f16, "isp_conv3x3_nhwc_f16"
    _need(x, BF16, "x")
    _need(Wt, BF16, "Wt")
    return _lib.lib().isp_conv3x3_nhwc_bf16, "isp_conv3x3_nhwc_bf16"


def conv3x3(x, Wt, bias=None, act="relu", out_dtype=None):
    """x [B,H,W,C] bf16 (or f16) NHWC, Wt [N, 9*C] same dtype (ky,kx,c order) -> [B,H,W,N] (dtype of x, or f32)."""
    fn, name = _conv_entry(x, Wt)
    out_dtype = x.dtype if out_dtype is None else out_dtype
    B, H, W, C = x.shape
    N = Wt.shape[0]
    if Wt.shape[1] != 9 * C:
        raise IspError("conv3x3 weight must be [N, 9*C]")
    out = torch.empty(B, H, W, N, device=x.device, dtype=out_dtype)
    if out_dtype == x.dtype:
        kind = {None: _lib.EP_BIAS_BF16, "relu": _lib.EP_BIAS_RELU_BF16, "gelu": _lib.EP_BIAS_GELU_BF16}[act]
    elif out_dtype == torch.float32 and x.dtype == BF16:
        kind = _lib.EP_BIAS_F32
    else:
        raise IspError("conv3x3: output must have the operands' dtype (or f32 for bf16 operands)")
    ep = _epilogue(kind, out, N, bias)
    check(fn(_p(x), _p(Wt), B, H, W, C, N, ctypes.byref(ep), _stream()), name)
    return out


def conv3x3_relu_stats(x, Wt, bias=None):
    """relu(conv3x3(x) + bias) on IEEE-half operands that also returns the per-pixel partial sums of the stored values:
    (out [B,H,W,N] f16, stats f32 [slots, B*H*W, 2]) -- what ``linear_lnfold`` needs to run a LayerNorm + Linear on `out` as one
    GEMM (csrc/gemm.hip, EpBiasActStats)."""
    _need(x, F16, "x")
    fn, name = _conv_entry(x, Wt)
    B, H, W, C = x.shape
    N = Wt.shape[0]
    if Wt.shape[1] != 9 * C:
        raise IspError("conv3x3 weight must be [N, 9*C]")
    out = torch.empty(B, H, W, N, device=x.device, dtype=F16)
    stats = torch.empty(_lib.lib().isp_conv_stats_slots(N), B * H * W, 2, device=x.device, dtype=torch.float32)
    ep = _epilogue(_lib.EP_BIAS_RELU_STATS_BF16, out, N, bias)
    ep.out2 = stats.data_ptr()
    check(fn(_p(x), _p(Wt), B, H, W, C, N, ctypes.byref(ep), _stream()), name)
    return out, stats


def conv3x3_folded_affine(x, Wt, bias_full, taps):
    """conv3x3 + ReLU whose input carries a folded per-pixel affine map: `bias_full` = conv bias +
    sum of the 9 tap constants `taps` [9,N]; border pixels drop the taps outside the image."""
    fn, name = _conv_entry(x, Wt)
    _need(bias_full, torch.float32, "bias_full")
    _need(taps, torch.float32, "taps")
    B, H, W, C = x.shape
    N = Wt.shape[0]
    out = torch.empty(B, H, W, N, device=x.device, dtype=x.dtype)
    ep = _epilogue(_lib.EP_BIAS_TAPS_RELU_BF16, out, N, bias_full, pos=taps)
    ep.img_h, ep.img_w = H, W
    check(fn(_p(x), _p(Wt), B, H, W, C, N, ctypes.byref(ep), _stream()), name)
    return out


def conv3x3_relu_classifier(x, Wt, bias, wcls, bcls):
    """relu(conv3x3(x) + bias) . wcls + bcls without storing the conv output: x [B,H,W,C] bf16 ->
    logits [B,H,W] f32 (second head conv + BaseClassifierHead.classifier fused)."""
    fn, name = _conv_entry(x, Wt)
    _need(bias, torch.float32, "bias")
    _need(wcls, torch.float32, "wcls")
    B, H, W, C = x.shape
    N = Wt.shape[0]
    M = B * H * W
    slots = _lib.lib().isp_conv3x3_partial_slots(N)
    partial = torch.empty(slots, M, device=x.device, dtype=torch.float32)
    ep = _epilogue(_lib.EP_RELU_DOT_PARTIAL_F32, partial, N, bias, gamma=wcls)
    check(fn(_p(x), _p(Wt), B, H, W, C, N, ctypes.byref(ep), _stream()), name)
    out = torch.empty(B, H, W, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_sum_partials_f32(_p(partial), _p(out), M, slots, float(bcls), _stream()), "isp_sum_partials_f32")
    return out


_DTYPE_CODE = {torch.float32: _lib.ISP_F32, BF16: _lib.ISP_BF16, torch.float16: _lib.ISP_F16}


def conv3x3_of_bilinear_supported(h, w, H, W, N, z_dtype=F16):
    """Whether ``conv3x3_of_bilinear`` has a kernel for this geometry (up-scaling by ~5.7 or more, N in whole channel blocks)."""
    return bool(_lib.lib().isp_conv3x3_of_bilinear_supported(h, w, H, W, N, _DTYPE_CODE[z_dtype]))


def conv3x3_of_bilinear_blend(z, bias, B, h, w, H, W, N, relu=True, out_dtype=F16):
    """Blend half of the first head conv taken through the bilinear resize (csrc/conv_bilinear.hip): z [B*h*w, 9*N] half (or
    f32), column t*N + n = (x W_t^T)[n] -> act(conv3x3(bilinear_ac(x, (H, W))) + bias) as [B,H,W,N] in ``out_dtype``."""
    if z.dtype not in (F16, torch.float32):
        raise IspError("conv3x3_of_bilinear_blend: z must be IEEE half or f32")
    _need(z, z.dtype, "z")
    if z.shape != (B * h * w, 9 * N):
        raise IspError(f"conv3x3_of_bilinear_blend: z must be [{B * h * w}, {9 * N}], got {tuple(z.shape)}")
    if bias is not None:
        _need(bias, torch.float32, "bias")
    out = torch.empty(B, H, W, N, device=z.device, dtype=out_dtype)
    check(_lib.lib().isp_conv3x3_of_bilinear_blend(_p(z), _DTYPE_CODE[z.dtype], _p(bias), _p(out), _DTYPE_CODE[out_dtype], B, h, w,
                                                   H, W, N, int(relu), _stream()), "isp_conv3x3_of_bilinear_blend")
    return out



def conv3x3_of_bilinear_blend_bwd(g, B, h, w, H, W, N):
    """Adjoint of ``conv3x3_of_bilinear_blend`` w.r.t. the tap planes (training): g [B,H,W,N] bf16 pre-activation gradient
    (ReLU mask applied) -> dz [B*h*w, 9*N] bf16."""
    _need(g, BF16, "g")
    if tuple(g.shape) != (B, H, W, N):
        raise IspError(f"conv3x3_of_bilinear_blend_bwd: g must be [{B}, {H}, {W}, {N}], got {tuple(g.shape)}")
    need = _lib.lib().isp_conv3x3_of_bilinear_bwd_workspace_bytes(B, h, W, N)
    if need < 0:
        check(int(need), "isp_conv3x3_of_bilinear_bwd_workspace_bytes")
    ws = torch.empty(need, device=g.device, dtype=torch.uint8)
    dz = torch.empty(B * h * w, 9 * N, device=g.device, dtype=BF16)
    check(_lib.lib().isp_conv3x3_of_bilinear_blend_bwd(_p(g), _p(dz), _p(ws), B, h, w, H, W, N, _stream()),
          "isp_conv3x3_of_bilinear_blend_bwd")
    return dz


def layernorm(x, gamma, beta, eps, out_dtype=BF16, group_out=0, skip=0, rows_out=None, D=None, ld_out=None):
    """LayerNorm over the first D columns of a [rows, ld] f32/bf16/f16 tensor (D defaults to ld); the
    output (f32, bf16, or f16) has row stride ld_out (default D) with columns [D, ld_out) zero-filled."""
    if x.dtype not in (torch.float32, BF16, F16):
        raise IspError("layernorm input must be f32, bf16 or f16")
    _need(x, x.dtype, "x")
    ld_in = x.shape[-1]
    D = ld_in if D is None else D
    ld_out = D if ld_out is None else ld_out
    rows = x.numel() // ld_in if rows_out is None else rows_out
    out = torch.empty(rows, ld_out, device=x.device, dtype=out_dtype)
    check(_lib.lib().isp_layernorm_fwd(_p(x), _p(out), _p(gamma), _p(beta), rows, D, float(eps),
                                       _DTYPE_CODE[x.dtype], _DTYPE_CODE[out_dtype],
                                       group_out, skip, ld_in, ld_out, _stream()), "isp_layernorm_fwd")
    return out


ATTENTION_LOGIT2_SCALE = 64 ** -0.5 * 1.4426950408889634  # what `q_logit2` expects to be folded into q (head_dim 64)


def attention_packed_qkv(qkv, B, L, heads, scale, q_logit2=False):
    """qkv [B*L, 3*heads*64] bf16 laid out (3, heads, 64) per token (attention.py:56-60).  q_logit2: the q third
    already carries scale * log2(e) (folded into the qkv weights, see DINOv2 featurizer): `scale` is then unused."""
    half = qkv.dtype == F16  # (the trunk's half-precision inference stream: q_logit2 form only)
    _need(qkv, F16 if half else BF16, "qkv")
    if half and not q_logit2:
        raise IspError("f16 qkv needs q_logit2=True")
    D = heads * 64
    out = torch.empty(B * L, D, device=qkv.device, dtype=qkv.dtype)
    base = qkv.data_ptr()
    q, k, v = (ctypes.c_void_p(base + i * D * 2) for i in range(3))
    strides = (L * 3 * D, 3 * D, 64, L * 3 * D, 3 * D, 64, L * D, D, 64)
    if q_logit2:
        fn, name = ((_lib.lib().isp_attention_fwd_logit2_f16, "isp_attention_fwd_logit2_f16") if half
                    else (_lib.lib().isp_attention_fwd_logit2, "isp_attention_fwd_logit2"))
        check(fn(q, k, v, _p(out), B, heads, L, L, 64, *strides, _stream()), name)
    else:
        check(_lib.lib().isp_attention_fwd(q, k, v, _p(out), B, heads, L, L, 64, *strides, float(scale), _stream()),
              "isp_attention_fwd")
    return out


def _stat_ld(L):
    return (L + 63) // 64 * 64


def attention_packed_qkv_lse(qkv, B, L, heads, scale):
    """attention_packed_qkv + the base-2 log-sum-exp [B*heads, round_up(L,64)] f32 the backward needs."""
    _need(qkv, BF16, "qkv")
    D = heads * 64
    out = torch.empty(B * L, D, device=qkv.device, dtype=BF16)
    lse = torch.zeros(B * heads, _stat_ld(L), device=qkv.device, dtype=torch.float32)
    base = qkv.data_ptr()
    q, k, v = (ctypes.c_void_p(base + i * D * 2) for i in range(3))
    check(_lib.lib().isp_attention_fwd_lse(q, k, v, _p(out), _p(lse), lse.shape[1], B, heads, L, L, 64,
                                           L * 3 * D, 3 * D, 64, L * 3 * D, 3 * D, 64, L * D, D, 64,
                                           float(scale), _stream()), "isp_attention_fwd_lse")
    return out, lse


def attention_packed_qkv_bwd(qkv, out, dout, lse, B, L, heads, scale):
    """Gradient of attention_packed_qkv w.r.t. the packed qkv: returns [B*L, 3*heads*64] bf16."""
    for t, n in ((qkv, "qkv"), (out, "out"), (dout, "dout")):
        _need(t, BF16, n)
    _need(lse, torch.float32, "lse")
    D = heads * 64
    if tuple(lse.shape) != (B * heads, _stat_ld(L)) or qkv.shape != (B * L, 3 * D) or out.shape != dout.shape:
        raise IspError("attention backward: shape mismatch")
    dqkv = torch.empty_like(qkv)
    delta = torch.zeros_like(lse)
    base, gbase = qkv.data_ptr(), dqkv.data_ptr()
    q, k, v = (ctypes.c_void_p(base + i * D * 2) for i in range(3))
    dq, dk, dv = (ctypes.c_void_p(gbase + i * D * 2) for i in range(3))
    check(_lib.lib().isp_attention_bwd(q, k, v, _p(out), _p(dout), _p(lse), _p(delta), lse.shape[1], dq, dk, dv,
                                       B, heads, L, L, 64, L * 3 * D, 3 * D, 64, L * 3 * D, 3 * D, 64, L * D, D, 64,
                                       float(scale), None, _stream()), "isp_attention_bwd")
    return dqkv


def resize_bilinear_nchw_f32_bwd(gout, h, w):
    """Adjoint of resize_bilinear_nchw_f32: gout [B,C,H,W] f32 -> [B,C,h,w] f32."""
    _need(gout, torch.float32, "gout")
    B, C, H, W = gout.shape
    gin = torch.empty(B, C, h, w, device=gout.device, dtype=torch.float32)
    check(_lib.lib().isp_resize_bilinear_ac_nchw_f32_bwd(_p(gout), _p(gin), B * C, h, w, H, W, _stream()),
          "isp_resize_bilinear_ac_nchw_f32_bwd")
    return gin


def layernorm_bwd(x, gy, gamma, eps, gx=None, group_out=0, skip=0, want_bf16=True, D=None):
    """gx (+)= d LayerNorm(x)/dx applied to gy.  x f32/bf16 [rows, ld_x] normalised over its first D columns
    (default all); gy bf16 [rows_out, ld_gy]; gx f32 [rows, ld_x] is accumulated into when given, else created
    (padding columns zero).  Returns (gx, bf16 copy of gx [rows, ld_x] or None)."""
    if x.dtype not in (torch.float32, BF16):
        raise IspError("layernorm_bwd input must be f32 or bf16")
    _need(x, x.dtype, "x")
    _need(gy, BF16, "gy", contiguous=False)
    rows, ld = x.shape
    D = ld if D is None else D
    accumulate = gx is not None
    if gx is None:
        gx = torch.empty(rows, ld, device=x.device, dtype=torch.float32)
    _need(gx, torch.float32, "gx")
    rows_out = rows if group_out == 0 else rows // (group_out + skip) * group_out
    if gy.dim() != 2 or gy.shape[0] != rows_out or gy.shape[1] < D or gy.stride(1) != 1 or gx.shape != (rows, ld):
        raise IspError("layernorm_bwd: shape mismatch")
    g16 = torch.empty(rows, ld, device=x.device, dtype=BF16) if want_bf16 else None
    check(_lib.lib().isp_layernorm_bwd(_p(x), _lib.ISP_F32 if x.dtype == torch.float32 else _lib.ISP_BF16, ld,
                                       _p(gy), gy.stride(0), _p(gamma), _p(gx), ld,
                                       _p(g16) if g16 is not None else None, ld, rows, D,
                                       float(eps), group_out, skip, int(accumulate), _stream()), "isp_layernorm_bwd")
    return gx, g16


def layernorm_wgrad(x, gy, eps, D=None):
    """(dgamma, dbeta) [D] f32 of LayerNorm over the first D columns of x [rows, ld] (f32 / bf16) given gy bf16."""
    if x.dtype not in (torch.float32, BF16):
        raise IspError("layernorm_wgrad input must be f32 or bf16")
    _need(x, x.dtype, "x")
    _need(gy, BF16, "gy", contiguous=False)
    rows, ld = x.shape
    D = ld if D is None else D
    if gy.dim() != 2 or gy.shape[0] != rows or gy.shape[1] < D or gy.stride(1) != 1:
        raise IspError("layernorm_wgrad: shape mismatch")
    dg = torch.zeros(D, device=x.device, dtype=torch.float32)
    db = torch.zeros(D, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_layernorm_wgrad(_p(x), _lib.ISP_F32 if x.dtype == torch.float32 else _lib.ISP_BF16, ld, _p(gy),
                                         gy.stride(0), _p(dg), _p(db), rows, D, float(eps), _stream()), "isp_layernorm_wgrad")
    return dg, db


def linear_wgrad(g, a, want_bias=True):
    """Weight (and bias) gradient of y = a W^T + b: dW [N,K] = g^T a (pixel-reduction GEMM), db [N] = column sums of g.
    g [M,N], a [M,K] bf16."""
    _need(g, BF16, "g")
    _need(a, BF16, "a")
    N, K = g.shape[1], a.shape[1]
    dw = torch.zeros(N, K, device=g.device, dtype=torch.float32)
    tn_gemm_atomic(g, a, dw)
    db = None
    if want_bias:
        ones = torch.ones(g.shape[0], 8, device=g.device, dtype=BF16)
        acc = torch.zeros(N, 8, device=g.device, dtype=torch.float32)
        tn_gemm_atomic(g, ones, acc)
        db = acc[:, 0].contiguous()
    return dw, db


def attention_bwd(q, k, v, out, dout, lse, scale, want_dq=True):
    """Backward of attention(): q/out/dout [B,Lq,H,hd], k/v [B,Lk,H,hd] bf16 (hd 64 or 128), lse from
    attention_lse().  Returns (dq or None, dk, dv) shaped like q, k, v (contiguous)."""
    hd = q.shape[3]
    for t, n in ((q, "q"), (k, "k"), (v, "v"), (out, "out"), (dout, "dout")):
        _need(t, BF16, n, contiguous=False)
        if t.stride(3) != 1 or t.shape[3] != hd:
            raise IspError(f"{n}: unit last-dim stride and equal head_dim required")
    if k.stride() != v.stride() or out.stride() != dout.stride():
        raise IspError("k/v and out/dout must share strides")
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    if tuple(lse.shape) != (B * H, _stat_ld(Lq)):
        raise IspError("lse shape mismatch")
    dq = torch.empty(B, Lq, H, hd, device=q.device, dtype=BF16) if want_dq else None
    dk = torch.empty(B, Lk, H, hd, device=q.device, dtype=BF16)
    dv = torch.empty_like(dk)
    if want_dq and dq.stride() != q.stride() or dk.stride() != k.stride():
        raise IspError("attention_bwd expects contiguous q, k, v (gradients share their strides)")
    delta = torch.zeros_like(lse)
    # few keys, many queries (cross-attention): let the kernel split the query range over blocks (fp32 dK/dV partials)
    split = torch.empty(2 * B * Lk * H * hd, device=q.device, dtype=torch.float32) if Lq >= 64 * Lk // 4 and Lk <= 4096 else None
    check(_lib.lib().isp_attention_bwd(_p(q), _p(k), _p(v), _p(out), _p(dout), _p(lse), _p(delta), lse.shape[1],
                                       _p(dq) if dq is not None else None, _p(dk), _p(dv), B, H, Lq, Lk, hd,
                                       q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                       out.stride(0), out.stride(1), out.stride(2), float(scale),
                                       _p(split) if split is not None else None, _stream()),
          "isp_attention_bwd")
    return dq, dk, dv


def attention_lse(q, k, v, scale):
    """attention() that also returns the base-2 log-sum-exp [B*H, round_up(Lq,64)] the backward needs."""
    hd = q.shape[3]
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, BF16, n, contiguous=False)
        if t.stride(3) != 1 or t.shape[3] != hd or hd not in (64, 128, 256):
            raise IspError(f"{n}: head_dim must be 64, 128 or 256 with unit stride")
    if k.stride() != v.stride():
        raise IspError("k and v must share strides")
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    out = torch.empty(B, Lq, H, hd, device=q.device, dtype=BF16)
    lse = torch.zeros(B * H, _stat_ld(Lq), device=q.device, dtype=torch.float32)
    check(_lib.lib().isp_attention_fwd_lse(_p(q), _p(k), _p(v), _p(out), _p(lse), lse.shape[1], B, H, Lq, Lk, hd,
                                           q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1), k.stride(2),
                                           out.stride(0), out.stride(1), out.stride(2), float(scale), _stream()),
          "isp_attention_fwd_lse")
    return out, lse


def attention(q, k, v, scale, q_logit2=False):
    """q [B,Lq,H,hd], k/v [B,Lk,H,hd] bf16 or f16, hd in {64,128,256} (any strides with unit last-dim stride).
    q_logit2: q already carries scale * log2(e) (``ATTENTION_LOGIT2_SCALE`` at head_dim 64; LoftUp folds its own into the
    query projection) -- the form the software-pipelined kernel takes; ``scale`` is then unused."""
    hd = q.shape[3]
    half = q.dtype == F16  # (LoftUp's inference stream)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, q.dtype if half else BF16, n, contiguous=False)
        if t.stride(3) != 1 or t.shape[3] != hd or hd not in (64, 128, 256):
            raise IspError(f"{n}: head_dim must be 64, 128 or 256 with unit stride")
    if k.stride() != v.stride():
        raise IspError("k and v must share strides")
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    out = torch.empty(B, Lq, H, hd, device=q.device, dtype=q.dtype)
    if q_logit2:
        fn, name = ((_lib.lib().isp_attention_fwd_logit2_f16, "isp_attention_fwd_logit2_f16") if half
                    else (_lib.lib().isp_attention_fwd_logit2, "isp_attention_fwd_logit2"))
        check(fn(_p(q), _p(k), _p(v), _p(out), B, H, Lq, Lk, hd, q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1),
                 k.stride(2), out.stride(0), out.stride(1), out.stride(2), _stream()), name)
        return out
    fn, name = (_lib.lib().isp_attention_fwd_f16, "isp_attention_fwd_f16") if half else (_lib.lib().isp_attention_fwd, "isp_attention_fwd")
    check(fn(_p(q), _p(k), _p(v), _p(out), B, H, Lq, Lk, hd, q.stride(0), q.stride(1), q.stride(2), k.stride(0), k.stride(1),
             k.stride(2), out.stride(0), out.stride(1), out.stride(2), float(scale), _stream()), name)
    return out


def attention_pipe(q, k, v):
    """The software-pipelined forward (csrc/attention_pipe.hip; off by default in the dispatch, ISEGPROBE_ATT_PIPE=1) through
    its own entry point: q [B,Lq,H,hd] carrying scale * log2(e), k/v [B,Lk,H,hd], bf16 or f16, hd 64 or 128, Lk >= 128."""
    hd = q.shape[3]
    half = q.dtype == F16
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _need(t, q.dtype if half else BF16, n, contiguous=False)
        if t.stride(3) != 1 or t.shape[3] != hd:
            raise IspError(f"{n}: unit last-dim stride and equal head_dim required")
    if k.stride() != v.stride():
        raise IspError("k and v must share strides")
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    out = torch.empty(B, Lq, H, hd, device=q.device, dtype=q.dtype)
    check(_lib.lib().isp_attention_fwd_pipe(_p(q), _p(k), _p(v), _p(out), B, H, Lq, Lk, hd, q.stride(0), q.stride(1), q.stride(2),
                                           k.stride(0), k.stride(1), k.stride(2), out.stride(0), out.stride(1), out.stride(2),
                                           int(half), _stream()), "isp_attention_fwd_pipe")
    return out


RESIZE_MODES = {"nearest": 0, "bilinear": 1, "bicubic": 2}


def resize_nhwc(x, H, W, mode):
    """NHWC bf16 resize: 'nearest' | 'bilinear' (align_corners=True) | 'bicubic' (align_corners=False)."""
    _need(x, BF16, "x")
    B, h, w, C = x.shape
    out = torch.empty(B, H, W, C, device=x.device, dtype=BF16)
    check(_lib.lib().isp_resize_nhwc_bf16(_p(x), _p(out), B, h, w, H, W, C, RESIZE_MODES[mode], _stream()),
          "isp_resize_nhwc_bf16")
    return out


def resize_bilinear_nhwc(x, H, W):
    return resize_nhwc(x, H, W, "bilinear")


def token_add_(x, add, B, T, has_cls):
    """x[b,(cls)+t,:] += add[b,t,:] in place; x/add f32 or bf16."""
    _need(x, x.dtype, "x")
    add = _need(add.contiguous(), add.dtype, "add")
    D = x.shape[-1]
    dt = {torch.float32: _lib.ISP_F32, BF16: _lib.ISP_BF16}
    check(_lib.lib().isp_token_add_fwd(_p(x), dt[x.dtype], _p(add), dt[add.dtype], B, T, D, int(has_cls), _stream()),
          "isp_token_add_fwd")
    return x


def resize_bilinear_nchw_f32(x, H, W):
    """x [..., h, w] f32 whose last two dims are contiguous; leading dims must be collapsible."""
    _need(x, torch.float32, "x")
    h, w = x.shape[-2:]
    planes = x.numel() // (h * w)
    out = torch.empty(*x.shape[:-2], H, W, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_resize_bilinear_ac_nchw_f32(_p(x), _p(out), planes, h, w, H, W, h * w, _stream()),
          "isp_resize_bilinear_ac_nchw_f32")
    return out


def classifier(x, weight, bias):
    """x [..., C] bf16 NHWC, weight [C] f32, bias python float -> [...] f32."""
    _need(x, BF16, "x")
    _need(weight, torch.float32, "weight")
    C = x.shape[-1]
    M = x.numel() // C
    out = torch.empty(x.shape[:-1], device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_classifier_fwd(_p(x), _p(weight), float(bias), _p(out), M, C, _stream()), "isp_classifier_fwd")
    return out


def nhwc_bf16_to_nchw_f32(x):
    _need(x, BF16, "x")
    B, H, W, C = x.shape
    out = torch.empty(B, C, H, W, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_nhwc_bf16_to_nchw_f32(_p(x), _p(out), B, C, H * W, _stream()), "isp_nhwc_bf16_to_nchw_f32")
    return out


def nchw_f32_to_nhwc_bf16(x):
    """x [B,C,H,W] f32 view with collapsible (H,W); returns NHWC bf16."""
    _need(x, torch.float32, "x", contiguous=False)
    B, C, H, W = x.shape
    if W > 1 and x.stride(2) != x.stride(3) * W:
        x = x.contiguous()
    out = torch.empty(B, H, W, C, device=x.device, dtype=BF16)
    check(_lib.lib().isp_nchw_f32_to_nhwc_bf16(_p(x), _p(out), B, C, H * W, x.stride(0), x.stride(1), x.stride(3),
                                               _stream()), "isp_nchw_f32_to_nhwc_bf16")
    return out


# ---------------------------------------------------------------------------------- JBU
def adaptive_avg_pool(x, OH, OW):
    """F.adaptive_avg_pool2d on contiguous fp32 [..., H, W]."""
    _need(x, torch.float32, "x")
    H, W = x.shape[-2:]
    out = torch.empty(*x.shape[:-2], OH, OW, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_adaptive_avg_pool_nchw_f32(_p(x), _p(out), x.numel() // (H * W), H, W, OH, OW, _stream()),
          "isp_adaptive_avg_pool_nchw_f32")
    return out


def _drop(drop, B, n):
    """Optional Dropout2d multipliers [B, n] f32 (0 or 1/(1-p)) -> pointer (or NULL)."""
    if drop is None:
        return None
    _need(drop, torch.float32, "drop")
    if tuple(drop.shape) != (B, n):
        raise IspError(f"dropout multipliers must be [{B}, {n}]")
    return _p(drop)


def jbu_range_proj(g, w0, b0, w3, b3, exact=False, drop=None):
    """g [B,3,GH,GW] f32 -> proj [B,GH,GW,32].  ``exact``: both layers in fp32 with the erf GELU, f32 output (the fp32
    checking mode); default: second layer on f16 MFMA, IEEE-half output (what jbu_kernels stages anyway)."""
    _need(g, torch.float32, "guidance")
    B, _, GH, GW = g.shape
    proj = torch.empty(B, GH, GW, 32, device=g.device, dtype=torch.float32 if exact else F16)
    check(_lib.lib().isp_jbu_range_proj(_p(g), _p(proj), _p(w0), _p(b0), _p(w3), _p(b3), B, GH, GW, int(bool(exact)),
                                        _drop(drop, B, 32), _stream()), "isp_jbu_range_proj")
    return proj


_JBU_TABLES = {}


def _cubic(t, A=-0.75):
    x0, x1, x2, x3 = t + 1.0, t, 1.0 - t, 2.0 - t
    return (((A * x0 - 5 * A) * x0 + 8 * A) * x0 - 4 * A, ((A + 2) * x1 - (A + 3)) * x1 * x1 + 1,
            ((A + 2) * x2 - (A + 3)) * x2 * x2 + 1, ((A * x3 - 5 * A) * x3 + 8 * A) * x3 - 4 * A)


def jbu_tables(G, device):
    """Size-only tables of the composite (reflect-pad o bicubic-x2) operator for an output extent G:
    rows [G,7,8] (window-relative) and cols [G,7,16] (circular slot = src col & 15).  See jbu.hip."""
    key = (G, str(device))
    if key not in _JBU_TABLES:
        import numpy as np
        rows = np.zeros((G, 7, 8), np.float32)
        cols = np.zeros((G, 7, 16), np.float32)
        for y in range(G):
            base = ((y - 4) >> 1) - 1
            for t in range(7):
                q = y + t - 3
                q = -q if q < 0 else q
                q = 2 * (G - 1) - q if q >= G else q          # F.pad(mode="reflect")
                i0 = ((q - 1) >> 1) - 1                       # first of the 4 bicubic source rows
                w4 = _cubic(0.75 if q % 2 == 0 else 0.25)
                for a in range(4):
                    rows[y, t, i0 + a - base] += w4[a]
                    cols[y, t, (i0 + a) & 15] += w4[a]
        _JBU_TABLES[key] = (torch.from_numpy(rows).to(device), torch.from_numpy(cols).to(device))
    return _JBU_TABLES[key]




def to_f16(x):
    """bf16 -> f16 (exact within half's range): the stack's input conversion."""
    if x.dtype == F16:
        return x
    _need(x, BF16, "x")
    out = torch.empty(x.shape, device=x.device, dtype=F16)
    check(_lib.lib().isp_bf16_to_f16(_p(x), _p(out), x.numel(), _stream()), "isp_bf16_to_f16")
    return out


def _proj_is_half(proj):
    if proj.dtype not in (torch.float32, F16) or not proj.is_contiguous() or proj.shape[-1] != 32:
        raise IspError("proj must be a contiguous [B,GH,GW,32] f32 or f16 tensor")
    return int(proj.dtype == F16)


def jbu_kernels(proj, g, f0w, f0b, f3w, f3b, range_temp, sigma_spatial, drop=None):
    """-> composite kernels kc [B,GH,GW,8,16] f16 (see include/isegprobe_hip.h); f0w / f3w: f16 [64,64]."""
    _need(f0w, F16, "f0w")
    _need(f3w, F16, "f3w")
    B, GH, GW, _ = proj.shape
    bys = jbu_tables(GH, proj.device)[0]
    bxs = jbu_tables(GW, proj.device)[1]
    kc = torch.empty(B, GH, GW, 8, 16, device=proj.device, dtype=F16)
    check(_lib.lib().isp_jbu_kernels(_p(proj), _proj_is_half(proj), _p(g), _p(kc), _p(f0w), _p(f0b), _p(f3w), _p(f3b), _p(bys), _p(bxs),
                                     float(range_temp), float(sigma_spatial), B, GH, GW, _drop(drop, B, 64), _stream()),
          "isp_jbu_kernels")
    return kc


def jbu_kernels_resized(proj, g, f0w, f0b, f3w, f3b, range_temp, sigma_spatial, OH, OW, drop=None):
    """jbu_blend(jbu_kernels(...), OH, OW) in one launch -> kc9 [B,OH,OW,9,16] f16."""
    _need(f0w, F16, "f0w")
    _need(f3w, F16, "f3w")
    B, GH, GW, _ = proj.shape
    bys = jbu_tables(GH, proj.device)[0]
    bxs = jbu_tables(GW, proj.device)[1]
    kc9 = torch.empty(B, OH, OW, 9, 16, device=proj.device, dtype=F16)
    check(_lib.lib().isp_jbu_kernels_resized(_p(proj), _proj_is_half(proj), _p(g), _p(kc9), _p(f0w), _p(f0b), _p(f3w), _p(f3b), _p(bys), _p(bxs),
                                             float(range_temp), float(sigma_spatial), B, GH, GW, OH, OW, _drop(drop, B, 64),
                                             _stream()), "isp_jbu_kernels_resized")
    return kc9


def jbu_apply(src, kc, out_dtype=F16):
    """src [B,h,w,C] f16 NHWC (bf16 is converted, exactly), kc [B,2h,2w,8,16] f16 -> [B,2h,2w,C]; ``out_dtype``: f16 for a
    map that feeds the next stage, bf16 for the one that leaves the stack."""
    src = to_f16(src)
    _need(kc, F16, "kc")
    B, h, w, C = src.shape
    out = torch.empty(B, 2 * h, 2 * w, C, device=src.device, dtype=out_dtype)
    check(_lib.lib().isp_jbu_apply(_p(src), _p(kc), _p(out), B, h, w, C, int(out_dtype == BF16), _stream()), "isp_jbu_apply")
    return out


def jbu_blend(kc, OH, OW):
    """Stage records [B,GH,GW,8,16] -> records of the bilinearly resized grid [B,OH,OW,9,16] (OH*8 == GH*7)."""
    _need(kc, F16, "kc")
    B, GH, GW = kc.shape[:3]
    out = torch.empty(B, OH, OW, 9, 16, device=kc.device, dtype=F16)
    check(_lib.lib().isp_jbu_blend(_p(kc), _p(out), B, GH, GW, OH, OW, _stream()), "isp_jbu_blend")
    return out


def jbu_apply_resized(src, kc9, out_dtype=BF16):
    """src [B,h,w,C] f16 NHWC (bf16 is converted), kc9 [B,OH,OW,9,16] f16 (jbu_blend) -> [B,OH,OW,C] = resize(jbu_apply(src, kc));
    this is the stage that leaves the stack, hence bf16 by default."""
    src = to_f16(src)
    _need(kc9, F16, "kc9")
    B, h, w, C = src.shape
    OH, OW = kc9.shape[1:3]
    out = torch.empty(B, OH, OW, C, device=src.device, dtype=out_dtype)
    check(_lib.lib().isp_jbu_apply_resized(_p(src), _p(kc9), _p(out), B, h, w, OH, OW, C, int(out_dtype == BF16), _stream()),
          "isp_jbu_apply_resized")
    return out


def jbu_apply_bwd(gout, kc):
    """Adjoint of jbu_apply w.r.t. src: gout [B,2h,2w,C] bf16, kc [B,2h,2w,8,16] f16 -> [B,h,w,C] bf16."""
    _need(gout, BF16, "gout")
    _need(kc, F16, "kc")
    B, GH, GW, C = gout.shape
    if kc.shape != (B, GH, GW, 8, 16) or GH % 2 or GW % 2:
        raise IspError("jbu_apply_bwd: shape mismatch")
    gsrc = torch.empty(B, GH // 2, GW // 2, C, device=gout.device, dtype=BF16)
    check(_lib.lib().isp_jbu_apply_bwd(_p(gout), _p(kc), _p(gsrc), B, GH // 2, GW // 2, C, _stream()), "isp_jbu_apply_bwd")
    return gsrc


def linear_axpy_res(A, Wt, bias, res, alpha, out=None):
    """bf16 (or f16, all operands alike): res + alpha * (A Wt^T + bias); ``out`` (same shape / dtype, contiguous) to write into."""
    _need(res, A.dtype, "res")
    N = Wt.shape[0]
    if out is None:
        out = torch.empty(A.shape[0], N, device=A.device, dtype=A.dtype)
    else:
        _need(out, A.dtype, "out")
        if tuple(out.shape) != (A.shape[0], N):
            raise IspError("linear_axpy_res: out shape mismatch")
    gemm(A, Wt, _epilogue(_lib.EP_AXPY_RES_BF16, out, N, bias, res=res, alpha=alpha))
    return out


def linear_axpy_res_stats(A, Wt, bias, res, alpha):
    """linear_axpy_res on IEEE-half operands (N <= 448) that also returns the per-row partial sums of the stored values:
    (out [M,N] f16, stats f32 [slots, M, 2] = (sum, sum of squares) per wave column group) -- the row statistics a
    LayerNorm of `out` needs, handed to ``linear_lnfold`` instead of running that LayerNorm (csrc/gemm.hip, EpAxpyResStats)."""
    _need(A, F16, "A")
    _need(res, F16, "res")
    N = Wt.shape[0]
    M = A.shape[0]
    out = torch.empty(M, N, device=A.device, dtype=F16)
    stats = torch.empty(_lib.lib().isp_gemm_stats_slots(), M, 2, device=A.device, dtype=torch.float32)
    ep = _epilogue(_lib.EP_AXPY_RES_STATS_BF16, out, N, bias, res=res, alpha=alpha)
    ep.out2 = stats.data_ptr()
    gemm(A, Wt, ep)
    return out, stats


def linear_lnfold(x, stats, Wfold, ssum, bias, D, eps, act=None):
    """act(Linear(LayerNorm_D(x))) WITHOUT the LayerNorm pass: x [M, K] f16 raw rows, stats from ``linear_axpy_res_stats``
    (the producer of x), Wfold = W diag(gain) in f16, ssum[n] = sum_k Wfold[n, k], bias = c + W ln_bias; D = channels the
    LayerNorm runs over (the remaining columns of x are zero padding).  act: None | 'gelu'."""
    _need(x, F16, "x")
    _need(stats, torch.float32, "stats")
    if stats.shape[1] != x.shape[0] or stats.shape[2] != 2:
        raise IspError("linear_lnfold: stats must be [slots, M, 2]")
    N = Wfold.shape[0]
    out = torch.empty(x.shape[0], N, device=x.device, dtype=F16)
    ep = _epilogue({None: _lib.EP_LNFOLD_BF16, "gelu": _lib.EP_LNFOLD_GELU_BF16}[act], out, N, bias, gamma=ssum, res=stats, alpha=float(eps))
    ep.tokens_per_image = int(D)
    ep.img_h = int(stats.shape[0])
    gemm(x, Wfold, ep)
    return out


def linear_lnfold_layernorm(x, stats, Wfold, ssum, bias, D, eps, gain2, bias2, eps2):
    """LayerNorm_N(Linear(LayerNorm_D(x))) as ONE GEMM: ``linear_lnfold`` whose epilogue also runs the LayerNorm over the N
    output channels (N <= 512: a tile spans the row; gain2 / bias2 f32 [N]) -- LoftUp's tail (loftup/loftup.py:139-149)
    without the pass that re-reads and re-writes the [pixels, C] map."""
    _need(x, F16, "x")
    _need(stats, torch.float32, "stats")
    if stats.shape[1] != x.shape[0] or stats.shape[2] != 2:
        raise IspError("linear_lnfold_layernorm: stats must be [slots, M, 2]")
    N = Wfold.shape[0]
    if N > 512 or N % 4:
        raise IspError("linear_lnfold_layernorm: N must be a multiple of 4, at most 512")
    out = torch.empty(x.shape[0], N, device=x.device, dtype=F16)
    ep = _epilogue(_lib.EP_LNFOLD_LAYERNORM_BF16, out, N, bias, gamma=ssum, res=stats, alpha=float(eps))
    ep.tokens_per_image, ep.img_h = int(D), int(stats.shape[0])
    ep.pos, ep.out2, ep.alpha2 = _p(_need(gain2, torch.float32, "gain2")), _p(_need(bias2, torch.float32, "bias2")), float(eps2)
    gemm(x, Wfold, ep)
    return out


def fuse_flip_sigmoid(logits, with_flip):
    """logits [2n,1,H,W] (second half: mirrored image) -> sigmoid(0.5*(a + flip(b))) [n,1,H,W];
    with_flip=False: sigmoid(logits)."""
    logits = _need(logits.contiguous(), torch.float32, "logits")
    N, C, H, W = logits.shape
    if with_flip and N % 2:
        raise IspError("flip fusion needs an even batch")
    n = N // 2 if with_flip else N
    out = torch.empty(n, C, H, W, device=logits.device, dtype=torch.float32)
    check(_lib.lib().isp_fuse_flip_sigmoid(_p(logits), _p(out), n * C, H, W, int(with_flip), _stream()),
          "isp_fuse_flip_sigmoid")
    return out


# ---------------------------------------------------------------------------------- LoftUp
def minmax_nchw(x):
    """Per-channel (min, max) over batch and space of an NCHW f32 tensor -> [C,2] f32."""
    _need(x, torch.float32, "x")
    B, C, H, W = x.shape
    out = torch.empty(C, 2, device=x.device, dtype=torch.float32)
    ws = torch.empty(C * B * 64 * 2, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_minmax_nchw_f32(_p(x), _p(out), _p(ws), B, C, H * W, _stream()), "isp_minmax_nchw_f32")
    return out


def loftup_fourier_cn(image, mm, freqs, bias_sin, bias_cos, gamma, beta, ldo, eps=1e-5, out_dtype=BF16):
    """image [B,3,H,W] f32 -> ChannelNorm(Fourier features) [B,H,W,ldo] bf16 (or fp32), zero-padded channels."""
    _need(image, torch.float32, "image")
    B, _, H, W = image.shape
    out = torch.empty(B, H, W, ldo, device=image.device, dtype=out_dtype)
    fn = {BF16: _lib.lib().isp_loftup_fourier_cn, F16: _lib.lib().isp_loftup_fourier_cn_f16,
          torch.float32: _lib.lib().isp_loftup_fourier_cn_f32}[out_dtype]
    check(fn(_p(image), _p(mm), _p(freqs), _p(bias_sin), _p(bias_cos), _p(gamma), _p(beta), _p(out), B, H, W,
             freqs.numel(), ldo, float(eps), _stream()), "isp_loftup_fourier_cn")
    return out


# ---------------------------------------------------------------------------------- LiFT
def conv3x3_s2_c32(x, w, bias, relu=True):
    """3x3 stride-2 pad-1 conv to 32 channels (+ ReLU unless relu=False).  x: NCHW f32 [B,3,H,W] or NHWC bf16
    [B,H,W,32]; w [32,3,3,cin] f32 -> NHWC bf16 [B,ceil(H/2),ceil(W/2),32]."""
    nchw = x.dtype == torch.float32
    _need(x, torch.float32 if nchw else BF16, "x")
    if nchw:
        B, cin, H, W = x.shape
    else:
        B, H, W, cin = x.shape
    out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, 32, device=x.device, dtype=BF16)
    check(_lib.lib().isp_conv3x3_s2_c32(_p(x), int(nchw), cin, _p(w), _p(bias), _p(out), B, H, W, int(bool(relu)), _stream()),
          "isp_conv3x3_s2_c32")
    return out


def bn_train(x, gamma, beta, eps, relu=True, running=None, momentum=0.1):
    """Train-mode BatchNorm2d (+ReLU) of an NHWC bf16 map x [..., C] with batch statistics over all leading dims.
    gamma / beta: f32 [C] (zero on padded channels).  running = (running_mean, running_var) f32 [c_real] asks for the
    statistics a torch BatchNorm2d would hold after this forward (returned, not written in place).
    Returns (y bf16, sums f32 [2C] for bn_train_bwd, (new_mean, new_var) or None)."""
    _need(x, BF16, "x")
    C = x.shape[-1]
    M = x.numel() // C
    sums = torch.zeros(2 * C, device=x.device, dtype=torch.float32)
    check(_lib.lib().isp_bn_train_stats(_p(x), _p(sums), M, C, _stream()), "isp_bn_train_stats")
    y = torch.empty_like(x)
    new = None
    rm = rv = nm = nv = None
    c_real = 0
    if running is not None:
        rm, rv = (_need(t.contiguous(), torch.float32, "running") for t in running)
        c_real = rm.numel()
        nm, nv = torch.empty_like(rm), torch.empty_like(rv)
        new = (nm, nv)
    check(_lib.lib().isp_bn_train_apply(_p(x), _p(sums), _p(gamma), _p(beta), _p(y), M, C, float(eps), int(bool(relu)),
                                        _p(rm) if rm is not None else None, _p(rv) if rv is not None else None,
                                        _p(nm) if nm is not None else None, _p(nv) if nv is not None else None, c_real,
                                        float(momentum), _stream()), "isp_bn_train_apply")
    return y, sums, new


def bn_train_bwd(dy, x, y, sums, gamma, eps):
    """Backward of bn_train w.r.t. x: dy is the gradient behind the ReLU whose output is y (y=None: no ReLU)."""
    _need(dy, BF16, "dy")
    _need(x, BF16, "x")
    C = x.shape[-1]
    M = x.numel() // C
    gsums = torch.zeros(2 * C, device=x.device, dtype=torch.float32)
    dx = torch.empty_like(x)
    check(_lib.lib().isp_bn_train_bwd(_p(dy), _p(x), _p(y) if y is not None else None, _p(sums), _p(gamma), _p(gsums), _p(dx),
                                      M, C, float(eps), _stream()), "isp_bn_train_bwd")
    return dx


def adaptive_max_pool_nhwc(x, OH, OW):
    _need(x, BF16, "x")
    B, H, W, C = x.shape
    out = torch.empty(B, OH, OW, C, device=x.device, dtype=BF16)
    check(_lib.lib().isp_adaptive_max_pool_nhwc_bf16(_p(x), _p(out), B, H, W, OH, OW, C, _stream()),
          "isp_adaptive_max_pool_nhwc_bf16")
    return out


# ---------------------------------------------------------------------------------- backward
def tn_gemm_atomic(P, Q, out, shift=None, splits=None):
    """out[n, j] += sum_m P[m, n] * Q[m', j]   (P [M,N], Q [M,J] bf16 row-major; out f32, pre-zeroed).
    shift=(H, W, dy, dx): Q rows are pixels of an NHWC map read at (y+dy, x+dx), zero outside."""
    _need(P, BF16, "P", contiguous=False)
    _need(Q, BF16, "Q", contiguous=False)
    _need(out, torch.float32, "out", contiguous=False)
    M, N = P.shape
    J = Q.shape[1]
    if splits is None:
        tiles = ((N + 127) // 128) * ((J + 127) // 128)
        splits = max(1, min((M + 1023) // 1024, 2048 // tiles))
    H, W, dy, dx = shift if shift else (0, 0, 0, 0)
    check(_lib.lib().isp_tn_gemm_bf16_atomic(_p(P), P.stride(0), _p(Q), Q.stride(0), _p(out), out.stride(0), M, N, J,
                                             H, W, dy, dx, splits, _stream()), "isp_tn_gemm_bf16_atomic")
    return out


def conv3x3_wgrad(g, x):
    """Weight gradient of conv3x3 (stride 1, pad 1): g [B,H,W,N], x [B,H,W,C] bf16 NHWC -> [N, 9*C] f32 in the
    layout of conv.weight.permute(0,2,3,1) (all nine taps from one staged input patch per 8x8 pixel tile)."""
    _need(g, BF16, "g")
    _need(x, BF16, "x")
    B, H, W, N = g.shape
    C = x.shape[3]
    if x.shape[:3] != g.shape[:3]:
        raise IspError("conv3x3_wgrad: g and x must share [B,H,W]")
    dw = torch.zeros(N, 9 * C, device=g.device, dtype=torch.float32)
    check(_lib.lib().isp_conv3x3_wgrad_bf16_atomic(_p(g), _p(x), _p(dw), B, H, W, C, N, _stream()),
          "isp_conv3x3_wgrad_bf16_atomic")
    return dw


def relu_mask_colsum(dy, y, want_colsum=True):
    _need(dy, BF16, "dy")
    _need(y, BF16, "y")
    N = y.shape[-1]
    M = y.numel() // N
    g = torch.empty_like(dy)
    cs = torch.zeros(N, device=y.device, dtype=torch.float32) if want_colsum else None
    check(_lib.lib().isp_relu_mask_colsum(_p(dy), _p(y), _p(g), _p(cs), M, N, _stream()), "isp_relu_mask_colsum")
    return g, cs


def classifier_bwd(grad_logits, x, w, want_dx_colsum=False, relu_mask=True):
    """grad_logits [M] f32, x [M,C] bf16, w [C] f32 -> (dx bf16, dw [C], db [1]) (+ the column sums of dx, i.e. the
    bias gradient of the conv that produced x, when asked).  ``relu_mask``: x is a post-ReLU map and dx is masked by
    x > 0 (the producing layer's ReLU backward); pass False when x is a signed feature map."""
    grad_logits = _need(grad_logits.contiguous(), torch.float32, "grad_logits")
    _need(x, BF16, "x")
    C = x.shape[-1]
    M = x.numel() // C
    dx = torch.empty_like(x)
    dw = torch.zeros(C, device=x.device, dtype=torch.float32)
    db = torch.zeros(1, device=x.device, dtype=torch.float32)
    cs = torch.zeros(C, device=x.device, dtype=torch.float32) if want_dx_colsum else None
    check(_lib.lib().isp_classifier_bwd(_p(grad_logits), _p(x), _p(w), _p(dx), _p(dw), _p(db),
                                        _p(cs) if cs is not None else None, M, C, int(bool(relu_mask)), _stream()),
          "isp_classifier_bwd")
    return (dx, dw, db, cs) if want_dx_colsum else (dx, dw, db)


def resize_bilinear_nhwc_bwd(dout, h, w):
    _need(dout, BF16, "dout")
    B, H, W, C = dout.shape
    din = torch.empty(B, h, w, C, device=dout.device, dtype=BF16)
    check(_lib.lib().isp_resize_bilinear_ac_nhwc_bwd(_p(dout), _p(din), B, h, w, H, W, C, _stream()),
          "isp_resize_bilinear_ac_nhwc_bwd")
    return din


def threshold_u8(probs, thr):
    """uint8 mask = probs > thr (evaluation.py:74), probs f32 of any shape."""
    probs = _need(probs.contiguous(), torch.float32, "probs")
    mask = torch.empty(probs.shape, device=probs.device, dtype=torch.uint8)
    check(_lib.lib().isp_threshold_u8(_p(probs), _p(mask), float(thr), probs.numel(), _stream()), "isp_threshold_u8")
    return mask


def robot_click(pred, gt, not_ignore, not_clicked, workspace=None):
    """Next robot click + IoU counts for uint8 [H,W] device masks (see isp_robot_click).  Returns the int32[8]
    DEVICE record {is_positive, row, col, fn_max_d2, fp_max_d2, intersection, union, 0} and the workspace."""
    for t, n in ((pred, "pred"), (gt, "gt"), (not_clicked, "not_clicked")):
        _need(t, torch.uint8, n)
    if not_ignore is not None:
        _need(not_ignore, torch.uint8, "not_ignore")
    H, W = gt.shape
    if pred.shape != gt.shape or not_clicked.shape != gt.shape or (not_ignore is not None and not_ignore.shape != gt.shape):
        raise IspError("robot_click: mask shapes differ")
    need = _lib.lib().isp_robot_click_workspace_bytes(H, W)
    if need < 0:
        check(int(need), "isp_robot_click_workspace_bytes")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=gt.device, dtype=torch.uint8)
    out = torch.empty(8, device=gt.device, dtype=torch.int32)
    check(_lib.lib().isp_robot_click(_p(pred), _p(gt), _p(not_ignore) if not_ignore is not None else None,
                                     _p(not_clicked), H, W, _p(workspace), _p(out), _stream()), "isp_robot_click")
    return out, workspace


def next_points(pred, gt, points, click_indx, rand32, pred_thresh=0.49, workspace=None):
    """Device get_next_points (trainer.py:575-618): returns an updated CLONE of points [B,2P,3] and the workspace.
    pred / gt [B,1,H,W] f32 on the GPU, rand32 [B] int64/uint32-valued tensor (the uniform draws)."""
    pred = _need(pred.contiguous(), torch.float32, "pred")
    gt = _need(gt.float().contiguous(), torch.float32, "gt")
    points = _need(points.clone().contiguous(), torch.float32, "points")
    B, _, H, W = pred.shape
    if gt.shape != pred.shape or points.shape[0] != B or points.shape[2] != 3 or points.shape[1] % 2:
        raise IspError("next_points: shape mismatch")
    # low 32 bits of each draw; the kernel reinterprets the int32 storage as uint32
    r32 = (rand32.to(device=pred.device, dtype=torch.int64) & 0xffffffff).to(torch.int32).contiguous()
    need = _lib.lib().isp_next_points_workspace_bytes(B, H, W)
    if need < 0:
        check(int(need), "isp_next_points_workspace_bytes")
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, device=pred.device, dtype=torch.uint8)
    check(_lib.lib().isp_next_points(_p(pred), _p(gt), _p(points), _p(r32), B, H, W, points.shape[1] // 2,
                                     int(click_indx), float(pred_thresh), _p(workspace), _stream()), "isp_next_points")
    return points, workspace


# ---------------------------------------------------------------- fp32-accurate products (core/model/precise.py)
def split3(x, weights=False, act=None, scale=1.0, K=None):
    """fp32 [rows, K] (last-dim contiguous, any row stride) -> bf16 [rows, 3*Kpad]: [hi|hi|lo] (activations) or
    [hi|lo|hi] (weights=True); Kpad = K rounded up to 64.  act in (None, 'relu', 'gelu') and scale apply first."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 2 or x.stride(1) != 1:
        raise IspError("split3 expects a CUDA fp32 matrix with contiguous rows")
    rows = x.shape[0]
    K = x.shape[1] if K is None else K
    Kpad = (K + 63) // 64 * 64
    out = torch.empty(rows, 3 * Kpad, device=x.device, dtype=BF16)
    ld = x.stride(0) if rows > 1 else max(x.stride(0), x.shape[1])  # (torch reports any stride for a size-1 dimension)
    check(_lib.lib().isp_split_bf16x3(_p(x), ld, _p(out), rows, K, Kpad, int(weights),
                                      {None: 0, "relu": 1, "gelu": 2, "quick_gelu": 3}[act], float(scale), _stream()), "isp_split_bf16x3")
    return out


def attention_packed_qkv_f32(qkv, B, L, heads, scale):
    """softmax((q * scale) k^T) v per (batch, head) in exact fp32 from the packed fp32 qkv [B*L, 3*heads*64] -> fp32 [B*L, heads*64]
    (csrc/attention_f32.hip: the fp32-accurate checking mode's attention, one launch)."""
    _need(qkv, torch.float32, "qkv")
    D = heads * 64
    if qkv.shape != (B * L, 3 * D):
        raise IspError(f"attention_packed_qkv_f32: qkv must be [{B * L}, {3 * D}]")
    out = torch.empty(B * L, D, device=qkv.device, dtype=torch.float32)
    check(_lib.lib().isp_attention_packed_f32(_p(qkv), _p(out), B, L, heads, float(scale), _stream()), "isp_attention_packed_f32")
    return out


def softmax_rows_(x, cols):
    """In-place softmax over the first `cols` columns of each row of a contiguous fp32 matrix; the rest becomes 0."""
    _need(x, torch.float32, "x")
    check(_lib.lib().isp_softmax_rows_f32(_p(x), x.shape[0], cols, x.shape[1], _stream()), "isp_softmax_rows_f32")
    return x


def jbu_stage_f32(src, proj, small, f0w, f0b, f3w, f3b, range_temp, sigma_spatial):
    """One FeatUp-JBU stage in plain fp32: src [B,h,w,C] f32 NHWC, proj [B,2h,2w,32] f32 (jbu_range_proj of `small`),
    small [B,3,2h,2w] f32 -> [B,2h,2w,C] f32 (per-pixel kernels, bicubic x2, adaptive 7x7 conv)."""
    for t, n in ((src, "src"), (proj, "proj"), (small, "small"), (f0w, "f0w"), (f3w, "f3w")):
        _need(t, torch.float32, n)
    B, h, w, C = src.shape
    GH, GW = 2 * h, 2 * w
    k = torch.empty(B, GH, GW, 49, device=src.device, dtype=torch.float32)
    check(_lib.lib().isp_jbu_kernels_f32(_p(proj), _p(small), _p(k), _p(f0w), _p(f0b), _p(f3w), _p(f3b), float(range_temp),
                                         float(sigma_spatial), B, GH, GW, _stream()), "isp_jbu_kernels_f32")
    hr = torch.empty(B, GH, GW, C, device=src.device, dtype=torch.float32)
    check(_lib.lib().isp_bicubic_x2_nhwc_f32(_p(src), _p(hr), B, h, w, C, _stream()), "isp_bicubic_x2_nhwc_f32")
    out = torch.empty_like(hr)
    check(_lib.lib().isp_adaptive_conv7_nhwc_f32(_p(hr), _p(k), _p(out), B, GH, GW, C, _stream()), "isp_adaptive_conv7_nhwc_f32")
    return out
