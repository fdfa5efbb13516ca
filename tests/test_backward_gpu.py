"""GPU: backward kernels against torch autograd (fp32 reference of the same op)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    from isegprobe_amd import hip_ops
    return hip_ops


def rel(a, b):
    return (a.float() - b.float()).abs().max().item() / (b.float().abs().max().item() + 1e-12)


@pytest.mark.parametrize("M,N,J", [(64, 128, 128), (1000, 64, 640), (6272, 128, 192), (4096, 384, 384)])
def test_tn_gemm_dense(ops, M, N, J):
    torch.manual_seed(M)
    P = torch.randn(M, N, device="cuda").to(BF)
    Q = torch.randn(M, J, device="cuda").to(BF)
    out = torch.zeros(N, J, device="cuda")
    ops.tn_gemm_atomic(P, Q, out)
    ref = P.float().t() @ Q.float()
    assert rel(out, ref) < 5e-3


def test_tn_gemm_asymmetric_layout(ops):
    """P = shifted identity-like pattern against an asymmetric Q: catches transposed fragments."""
    M, N, J = 128, 128, 128
    P = torch.zeros(M, N, device="cuda")
    P[torch.arange(M), (torch.arange(M) * 7 + 3) % N] = 1.0
    Q = ((torch.arange(M * J, device="cuda").float().reshape(M, J) % 253) / 8).to(BF)
    out = torch.zeros(N, J, device="cuda")
    ops.tn_gemm_atomic(P.to(BF), Q, out)
    assert torch.equal(out, P.t() @ Q.float())


@pytest.mark.parametrize("B,H,W,C,N", [(2, 20, 24, 64, 64), (1, 37, 45, 128, 192), (2, 56, 56, 128, 128)])
def test_conv_wgrad_dgrad_vs_autograd(ops, B, H, W, C, N):
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W, device="cuda").to(BF).float().requires_grad_(True)
    w = (torch.randn(N, C, 3, 3, device="cuda") / math.sqrt(9 * C)).to(BF).float().requires_grad_(True)
    bias = torch.randn(N, device="cuda", requires_grad=True)
    y = F.relu(F.conv2d(x, w, bias, padding=1))
    gy = torch.randn_like(y).to(BF).float()
    y.backward(gy)
    # HIP: mask, bias grad, weight grad (9 tap GEMMs), data grad (conv with rotated weights)
    x_n = x.detach().permute(0, 2, 3, 1).contiguous().to(BF)
    y_n = y.detach().permute(0, 2, 3, 1).contiguous().to(BF)
    gy_n = gy.permute(0, 2, 3, 1).contiguous().to(BF)
    g, db = ops.relu_mask_colsum(gy_n, y_n)
    assert rel(db, bias.grad) < 5e-3
    dw = torch.zeros(N, 9 * C, device="cuda")
    M = B * H * W
    for t in range(9):
        ops.tn_gemm_atomic(g.view(M, N), x_n.view(M, C), dw[:, t * C:(t + 1) * C], shift=(H, W, t // 3 - 1, t % 3 - 1))
    dw_ref = w.grad.permute(0, 2, 3, 1).reshape(N, 9 * C)
    assert rel(dw, dw_ref) < 1e-2
    dw9 = ops.conv3x3_wgrad(g, x_n)  # all nine taps from one staged 10x10 patch per 8x8 tile
    assert rel(dw9, dw_ref) < 1e-2
    w_rot = w.detach().flip(2, 3).permute(1, 2, 3, 0).reshape(C, 9 * N).to(BF).contiguous()  # [C][ky][kx][N]
    dx = ops.conv3x3(g, w_rot, None, None)
    assert rel(dx.permute(0, 3, 1, 2), x.grad) < 1e-2


@pytest.mark.parametrize("B,H,W,C,N", [(1, 8, 8, 64, 128), (2, 19, 13, 72, 40), (1, 64, 72, 384, 384)])
def test_conv_wgrad_fused_taps_asymmetric(ops, B, H, W, C, N):
    """Structured (non-random) operands so that a transposed fragment, a wrong tap shift or a wrong tile
    origin gives an O(1) error: g is one-hot per pixel in a pixel-dependent channel, x a smooth ramp.
    Sizes include ragged spatial tiles (19x13) and channel counts that are not tile multiples (72, 40)."""
    ys, xs = torch.meshgrid(torch.arange(H, device="cuda"), torch.arange(W, device="cuda"), indexing="ij")
    g = torch.zeros(B, H, W, N, device="cuda")
    for b in range(B):
        g[b].view(H * W, N)[torch.arange(H * W, device="cuda"), ((ys * 7 + xs * 3 + b) % N).flatten()] = 1.0
    x = ((ys * 5 + xs * 11)[None, :, :, None] % 23 + torch.arange(C, device="cuda")[None, None, None, :] % 7 - 12).float() / 8
    x = x.expand(B, H, W, C).contiguous()
    dw = ops.conv3x3_wgrad(g.to(BF), x.to(BF))
    xr = x.permute(0, 3, 1, 2).contiguous()
    wr = torch.zeros(N, C, 3, 3, device="cuda", requires_grad=True)
    F.conv2d(xr, wr, padding=1).backward(g.permute(0, 3, 1, 2).contiguous())
    ref = wr.grad.permute(0, 2, 3, 1).reshape(N, 9 * C)
    assert rel(dw, ref) < 2e-3


@pytest.mark.parametrize("M,C", [(5000, 128), (777, 384), (4099, 24), (1000, 1024)])
def test_classifier_bwd(ops, M, C):
    """C = 384 leaves idle threads in the (row slot, 8 columns) mapping, M % 256 != 0 a partial block."""
    torch.manual_seed(C)
    x = F.relu(torch.randn(M, C, device="cuda")).to(BF)
    w = torch.randn(C, device="cuda")
    gl = torch.randn(M, device="cuda")
    dx, dw, db = ops.classifier_bwd(gl, x, w)
    ref_dx = (x.float() > 0) * gl[:, None] * w[None, :]
    assert rel(dx, ref_dx) < 1e-2
    assert rel(dw, (gl[:, None] * x.float()).sum(0)) < 1e-3
    assert abs(db.item() - gl.sum().item()) < 1e-2
    dx2, dw2, db2, cs = ops.classifier_bwd(gl, x, w, want_dx_colsum=True)
    assert torch.equal(dx2, dx)
    assert rel(cs, dx.float().sum(0)) < 1e-3
    # signed input (the "linear" head / num_layers = 0): no ReLU mask on dx
    xs = torch.randn(M, C, device="cuda").to(BF)
    dx3, dw3, db3 = ops.classifier_bwd(gl, xs, w, relu_mask=False)
    assert rel(dx3, gl[:, None] * w[None, :].expand(M, C)) < 1e-2
    assert rel(dw3, (gl[:, None] * xs.float()).sum(0)) < 1e-3


@pytest.mark.parametrize("M,N", [(5000, 128), (300, 384), (1025, 8)])
def test_relu_mask_colsum(ops, M, N):
    torch.manual_seed(N)
    dy = torch.randn(M, N, device="cuda").to(BF)
    y = F.relu(torch.randn(M, N, device="cuda")).to(BF)
    g, cs = ops.relu_mask_colsum(dy, y)
    ref = dy.float() * (y.float() > 0)
    assert torch.equal(g.float(), ref)
    assert rel(cs, ref.sum(0)) < 1e-3


@pytest.mark.parametrize("shape", [(2, 4, 5, 56, 70, 64), (1, 16, 16, 224, 224, 128)])
def test_bilinear_bwd(ops, shape):
    B, h, w, H, W, C = shape
    x = torch.randn(B, C, h, w, device="cuda", requires_grad=True)
    y = F.interpolate(x, (H, W), mode="bilinear", align_corners=True)
    gy = torch.randn_like(y).to(BF).float()
    y.backward(gy)
    din = ops.resize_bilinear_nhwc_bwd(gy.permute(0, 2, 3, 1).contiguous().to(BF), h, w)
    assert rel(din.permute(0, 3, 1, 2), x.grad) < 1e-2


# ------------------------------------------------------------------ frozen-ViT backward pieces
@pytest.mark.parametrize("B,L,heads", [(2, 257, 2), (1, 64, 1), (2, 1025, 6), (3, 130, 3)])
def test_attention_backward(ops, B, L, heads):
    """dQ/dK/dV of the fused attention against autograd of the materialised softmax (fp32 on the
    same bf16-rounded inputs).  L = 257 / 1025 / 130 exercise the partial last tile."""
    torch.manual_seed(L + heads)
    D = heads * 64
    scale = 64 ** -0.5
    qkv = torch.randn(B * L, 3 * D, device="cuda").to(BF)
    dout = torch.randn(B * L, D, device="cuda").to(BF)
    out, lse = ops.attention_packed_qkv_lse(qkv, B, L, heads, scale)
    assert torch.equal(out, ops.attention_packed_qkv(qkv, B, L, heads, scale))  # same forward
    dqkv = ops.attention_packed_qkv_bwd(qkv, out, dout, lse, B, L, heads, scale)

    ref_in = qkv.float().requires_grad_(True)
    q, k, v = ref_in.view(B, L, 3, heads, 64).permute(2, 0, 3, 1, 4)
    s = (q * scale) @ k.transpose(-2, -1)
    ref_lse = torch.logsumexp(s, -1) * math.log2(math.e)  # base 2
    o = (s.softmax(-1) @ v).transpose(1, 2).reshape(B * L, D)
    o.backward(dout.float())
    assert (lse[:, :L] - ref_lse.reshape(B * heads, L)).abs().max().item() < 2e-2
    g, r = dqkv.float().view(B * L, 3, D), ref_in.grad.view(B * L, 3, D)
    for i, name in enumerate("qkv"):
        err = (g[:, i] - r[:, i]).abs().max().item() / r[:, i].abs().max().item()
        assert err < 2e-2, (name, err)


def test_attention_backward_asymmetric(ops):
    """One-hot style probe: a dominant key per query makes P ~ a permutation, so transposed or
    mis-indexed fragments show up as O(1) errors rather than noise."""
    B, L, heads = 1, 192, 1
    torch.manual_seed(5)
    q = torch.randn(L, 64, device="cuda") * 0.05
    k = torch.randn(L, 64, device="cuda") * 0.05
    perm = (torch.arange(L, device="cuda") * 37 + 11) % L
    k[perm] += q * 60.0
    v = torch.randn(L, 64, device="cuda")
    qkv = torch.stack((q, k, v), 1).reshape(L, 192).to(BF)
    dout = torch.randn(L, 64, device="cuda").to(BF)
    out, lse = ops.attention_packed_qkv_lse(qkv, B, L, heads, 1.0)
    dqkv = ops.attention_packed_qkv_bwd(qkv, out, dout, lse, B, L, heads, 1.0).float().view(L, 3, 64)
    ref_in = qkv.float().requires_grad_(True)
    qq, kk, vv = ref_in.view(L, 3, 64).unbind(1)
    ((qq @ kk.t()).softmax(-1) @ vv).backward(dout.float())
    r = ref_in.grad.view(L, 3, 64)
    for i in range(3):
        assert (dqkv[:, i] - r[:, i]).abs().max().item() / r[:, i].abs().max().item() < 3e-2


@pytest.mark.parametrize("rows,D,group_out,skip", [(257 * 2, 384, 0, 0), (64, 128, 0, 0), (3 * 26, 768, 25, 1), (10, 1024, 0, 0)])
def test_layernorm_backward(ops, rows, D, group_out, skip):
    torch.manual_seed(rows)
    x = torch.randn(rows, D, device="cuda") * 2 + 0.5
    gamma = torch.randn(D, device="cuda")
    beta = torch.randn(D, device="cuda")
    rows_out = rows if group_out == 0 else rows // (group_out + skip) * group_out
    gy = torch.randn(rows_out, D, device="cuda").to(BF)
    xr = x.clone().requires_grad_(True)
    y = F.layer_norm(xr, (D,), gamma, beta, 1e-6)
    if group_out:
        y = y.view(-1, group_out + skip, D)[:, skip:].reshape(rows_out, D)
    y.backward(gy.float())
    gx, g16 = ops.layernorm_bwd(x, gy, gamma, 1e-6, group_out=group_out, skip=skip)
    assert rel(gx, xr.grad) < 1e-4
    assert rel(g16, xr.grad) < 1e-2
    base = torch.randn_like(x)
    acc, _ = ops.layernorm_bwd(x, gy, gamma, 1e-6, gx=base.clone(), group_out=group_out, skip=skip)
    assert rel(acc, base + xr.grad) < 1e-4


def test_gelu_save_and_dgelu_epilogues(ops):
    torch.manual_seed(3)
    M, K, N = 300, 128, 512
    A = torch.randn(M, K, device="cuda").to(BF)
    W = (torch.randn(N, K, device="cuda") / 8).to(BF)
    b = torch.randn(N, device="cuda")
    hid, pre = ops.linear_gelu_save(A, W, b)
    assert torch.equal(hid, ops.linear(A, W, b, "gelu"))       # identical to the inference epilogue
    assert torch.equal(pre, ops.linear(A, W, b, None))
    G = torch.randn(M, K, device="cuda").to(BF)
    out = ops.linear_mul_dgelu(G, W, pre)
    p = pre.float().requires_grad_(True)
    F.gelu(p).backward(G.float() @ W.float().t())
    assert rel(out, p.grad) < 1e-2


def test_logits_resize_backward(ops):
    g = torch.randn(3, 1, 56, 70, device="cuda")
    x = torch.randn(3, 1, 4, 5, device="cuda", requires_grad=True)
    F.interpolate(x, (56, 70), mode="bilinear", align_corners=True).backward(g)
    assert rel(ops.resize_bilinear_nchw_f32_bwd(g, 4, 5), x.grad) < 1e-5


@pytest.mark.parametrize("B,Lq,Lk,H,hd,real", [(2, 300, 130, 2, 128, 101), (1, 3136, 16, 4, 128, 37), (2, 200, 77, 3, 64, 64),
                                               (8, 1190, 1100, 4, 128, 101),   # 32 owner rows per wave (both launches)
                                               (2, 333, 150, 2, 256, 197),    # LoftUp(768): head_dim 197 padded to 256
                                               (2, 12544, 256, 4, 128, 101)])  # LoftUp at the 224^2 crop: split query range, fp32 dK/dV partials
def test_cross_attention_backward(ops, B, Lq, Lk, H, hd, real):
    """Lq != Lk, head_dim 128 with zero padding beyond `real` (LoftUp: 101 -> 128), dQ optional."""
    torch.manual_seed(Lq)
    def mk(L):
        t = torch.randn(B, L, H, hd, device="cuda")
        t[..., real:] = 0
        return t.to(BF)
    q, k, v, dout = mk(Lq), mk(Lk), mk(Lk), mk(Lq)
    scale = real ** -0.5
    out, lse = ops.attention_lse(q, k, v, scale)
    assert torch.equal(out, ops.attention(q, k, v, scale))
    dq, dk, dv = ops.attention_bwd(q, k, v, out, dout, lse, scale)
    none_dq, dk2, dv2 = ops.attention_bwd(q, k, v, out, dout, lse, scale, want_dq=False)
    assert none_dq is None
    if Lq < 16 * Lk:  # (the split-query launch adds fp32 partials with atomics: order-dependent rounding)
        assert torch.equal(dk, dk2) and torch.equal(dv, dv2)
    else:
        assert rel(dk, dk2) < 1e-2 and rel(dv, dv2) < 1e-2
    qf, kf, vf = (t.float().requires_grad_(True) for t in (q, k, v))
    p = ((qf.permute(0, 2, 1, 3) * scale) @ kf.permute(0, 2, 3, 1)).softmax(-1)
    (p @ vf.permute(0, 2, 1, 3)).permute(0, 2, 1, 3).backward(dout.float())
    for name, g, r in (("dq", dq, qf.grad), ("dk", dk, kf.grad), ("dv", dv, vf.grad)):
        assert rel(g, r) < 2e-2, name
        if real < hd:  # zero padding stays zero
            assert g[..., real:].abs().max().item() == 0, name


def test_layernorm_backward_bf16_padded(ops):
    """bf16 input with row stride > D (LoftUp's 404 channels in 448-wide rows), strided gy."""
    rows, D, ld = 333, 404, 448
    torch.manual_seed(1)
    x = torch.zeros(rows, ld, device="cuda")
    x[:, :D] = torch.randn(rows, D, device="cuda") * 1.5 + 0.3
    x = x.to(BF)
    gamma = torch.randn(D, device="cuda")
    gy_full = torch.randn(rows, ld + 64, device="cuda").to(BF)
    gy = gy_full[:, :ld]  # row stride ld + 64
    xr = x[:, :D].float().requires_grad_(True)
    F.layer_norm(xr, (D,), gamma, torch.zeros(D, device="cuda"), 1e-5).backward(gy[:, :D].float())
    gx, g16 = ops.layernorm_bwd(x, gy, gamma, 1e-5, D=D)
    assert rel(gx[:, :D], xr.grad) < 1e-4
    assert gx[:, D:].abs().max().item() == 0 and g16[:, D:].abs().max().item() == 0
    assert rel(g16[:, :D], xr.grad) < 1e-2


@pytest.mark.parametrize("B,h,w,C", [(2, 4, 5, 64), (1, 8, 8, 128), (1, 16, 12, 192)])
def test_jbu_apply_adjoint(ops, B, h, w, C):
    """<A x, g> == <x, A^T g> for the composite-kernel apply (borders included: clamped taps pile up on edge pixels),
    plus explicit columns of A (apply on a one-hot source) against the matching entries of A^T g."""
    torch.manual_seed(h * w)
    kc = torch.rand(B, 2 * h, 2 * w, 8, 16, device="cuda") / 8
    # the kernel format: of the 16 circular column slots only the 8 of the pixel's window [base_x, base_x + 8) are non-zero
    xs = torch.arange(2 * w, device="cuda")
    bx = ((xs - 4) >> 1) - 1
    live = torch.zeros(2 * w, 16, device="cuda")
    for rx in range(8):
        live[xs, (bx + rx) & 15] = 1
    kc = (kc * live[None, None, :, None, :]).to(torch.float16)  # records are IEEE half inside the JBU stack
    x = torch.randn(B, h, w, C, device="cuda").to(BF)
    g = torch.randn(B, 2 * h, 2 * w, C, device="cuda").to(BF)
    Ax = ops.jbu_apply(x, kc).float()
    ATg = ops.jbu_apply_bwd(g, kc).float()
    lhs, rhs = (Ax * g.float()).sum().item(), (x.float() * ATg).sum().item()
    assert abs(lhs - rhs) < 2e-2 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)
    for (sy, sx) in ((0, 0), (h - 1, w - 1), (h // 2, w // 2), (0, w // 2)):
        e = torch.zeros(B, h, w, C, device="cuda", dtype=BF)
        e[0, sy, sx, 3] = 1.0
        col = ops.jbu_apply(e, kc)[0, :, :, 3].float()            # column (sy, sx) of A for batch 0, channel 3
        want = (col * g[0, :, :, 3].float()).sum().item()
        assert abs(ATg[0, sy, sx, 3].item() - want) < 2e-2 * max(1.0, abs(want)), (sy, sx)
